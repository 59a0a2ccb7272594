"""Seeded synthetic weights: the generators live in captra_amd/synthetic.py; this module keeps the tests' import path."""
from captra_amd.synthetic import make_physical_state_dict, make_state_dict, shapes_of  # noqa: F401
