"""CPU oracle for the CAPTRA hot path — TEST INFRASTRUCTURE ONLY.

`oracle.ops` wraps the plain-C restatement (captra_oracle.c) with numpy; `oracle.model` restates
the network / track loop on top of it.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; captra_amd never does.
"""
