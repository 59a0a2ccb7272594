"""Seeded synthetic clouds: the generators live in captra_amd/synthetic.py (the product harnesses use them
too); this module keeps the tests' import path."""
from captra_amd.synthetic import (PHYSICAL_SETUPS, PHYSICAL_SETUPS_MORE, _rot_x, _rot_y, make_trajectory, s_arti, s_nocs, s_nocs_dup,  # noqa: F401
                                  s_uni)
