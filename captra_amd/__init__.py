"""captra_amd — MI355X (gfx950) implementation of CAPTRA's per-frame point-cloud hot path.

Layout: csrc/ (HIP kernels + the C ABI of include/captra_hip.h), pointnet2_cuda (drop-in for the
reference's pybind module), pointnet_lib/ (op layer), the network / pose / track-loop mirrors of
the reference's host code.  There is no CPU fallback anywhere in this package.
"""
import sys as _sys

__version__ = "0.1.0"


def install_as_pointnet2_cuda() -> None:
    """Make `import pointnet2_cuda` resolve to the HIP-backed module (for the reference's own
    network/models/pointnet_lib/pointnet2_utils.py:7)."""
    from . import pointnet2_cuda as _m
    _sys.modules["pointnet2_cuda"] = _m
