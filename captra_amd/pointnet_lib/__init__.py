"""Python op layer over the HIP kernels; mirrors the reference package network/models/pointnet_lib."""
from . import pointnet2_utils  # noqa: F401
