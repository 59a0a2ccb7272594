"""Rotation representations used on the tracking path.

Mirrors the used subset of the reference's pose_utils/rotations.py: `normalize_vector` (l.302-314),
`cross_product` (l.317-327), `compute_rotation_matrix_from_ortho6d` (l.330-343),
`compute_rotation_matrix_from_matrix` (Gram-Schmidt, l.356-372),
`compute_rotation_matrix_from_3d` (l.375-387), the quaternion helpers and `noisy_rot_matrix`
(l.278-287).  All functions are device-agnostic tensor algebra on tiny (B*P) batches.
"""
from __future__ import annotations

import torch

_TINY = 1e-8


def normalize_vector(v: torch.Tensor) -> torch.Tensor:
    """v (B,3) -> v/|v|, with (1,0,0) where |v| <= 1e-8."""
    mag = torch.linalg.vector_norm(v, dim=1)
    ok = (mag > _TINY).to(v.dtype).unsqueeze(1)
    unit = v / torch.clamp_min(mag, _TINY).unsqueeze(1)
    backup = torch.zeros_like(v)          # (1,0,0) built on device: no host->device copy, hipGraph-capturable
    backup[:, 0] = 1.0
    return unit * ok + backup * (1.0 - ok)


def cross_product(u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    return torch.stack((u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1],
                        u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2],
                        u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]), dim=1)


def compute_rotation_matrix_from_ortho6d(poses: torch.Tensor) -> torch.Tensor:
    """(B,6) -> (B,3,3) with columns x, y, z (x from the first triple, z = x × y_raw)."""
    x = normalize_vector(poses[:, 0:3])
    z = normalize_vector(cross_product(x, poses[:, 3:6]))
    y = cross_product(z, x)
    return torch.stack((x, y, z), dim=2)


def _proj(u: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    top = (u * a).sum(dim=1)
    bottom = torch.clamp_min((u * u).sum(dim=1), _TINY)
    return (top / bottom).unsqueeze(1) * u


def compute_rotation_matrix_from_matrix(matrices: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt on the COLUMNS of (B,3,3)."""
    a1, a2, a3 = matrices[:, :, 0], matrices[:, :, 1], matrices[:, :, 2]
    u1 = a1
    u2 = a2 - _proj(u1, a2)
    u3 = a3 - _proj(u1, a3) - _proj(u2, a3)
    return torch.stack((normalize_vector(u1), normalize_vector(u2), normalize_vector(u3)), dim=2)


def compute_rotation_matrix_from_3d(vec: torch.Tensor) -> torch.Tensor:
    """(B,3) y-axis direction -> (B,3,3) with columns x, y, z; z = e_x × y."""
    y = normalize_vector(vec)
    ex = torch.zeros_like(y)
    ex[:, 0] = 1.0
    z = normalize_vector(cross_product(ex, y))
    x = cross_product(y, z)
    return torch.stack((x, y, z), dim=2)


# ---- quaternions (w, x, y, z) ---------------------------------------------------------------------
def normalize(q: torch.Tensor) -> torch.Tensor:
    return q / q.norm(dim=-1, keepdim=True)


def unit_quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = torch.unbind(q, dim=-1)
    m = torch.stack((1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                     2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                     2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y), dim=-1)
    return m.reshape(m.shape[:-1] + (3, 3)).contiguous()


def matrix_to_unit_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    trace = torch.clamp_min(1 + matrix[..., 0, 0] + matrix[..., 1, 1] + matrix[..., 2, 2], 0.0)
    r = torch.sqrt(trace)
    s = 1.0 / (2 * r + 1e-7)
    q = torch.stack((0.5 * r,
                     (matrix[..., 2, 1] - matrix[..., 1, 2]) * s,
                     (matrix[..., 0, 2] - matrix[..., 2, 0]) * s,
                     (matrix[..., 1, 0] - matrix[..., 0, 1]) * s), dim=-1)
    return normalize(q)


def generate_random_quaternion(shape) -> torch.Tensor:
    """Drawn from the CPU generator, like the reference (rotations.py:271-275)."""
    assert shape[-1] == 4
    return normalize(torch.randn(shape))


def jitter_quaternion(q: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Rotate q by angle theta (…,1) about a random axis: q cos(θ/2) + q⊥ sin(θ/2)."""
    rnd = generate_random_quaternion(q.shape).to(q.device)
    dot = (q * rnd).sum(dim=-1, keepdim=True)
    q_orth = normalize(rnd - q * dot)
    return q * torch.cos(theta / 2) + q_orth * torch.sin(theta / 2)


def noisy_rot_matrix(matrix: torch.Tensor, rad: float, type: str = "normal") -> torch.Tensor:
    """Perturb rotations by |N(0,1)|*rad (normal) or U(0,1)*rad (uniform).  Random numbers come
    from the CPU generator so that a seeded run yields the same noise on every device."""
    shape = matrix[..., 0, 0].shape
    if type == "normal":
        theta = torch.abs(torch.randn(shape)) * rad
    elif type == "uniform":
        theta = torch.rand(shape) * rad
    else:
        raise ValueError(type)
    q = matrix_to_unit_quaternion(matrix)
    return unit_quaternion_to_matrix(jitter_quaternion(q, theta.to(matrix.device).unsqueeze(-1)))


# ---- axis-angle / rotation-vector conversions (reference rotations.py:109-155): used by the exponential-map metrics -----
def axis_theta_to_quater(axis: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """unit axis (…,3), angle (…) -> unit quaternion (w, x, y, z)."""
    half = theta / 2.0
    return normalize(torch.cat([torch.cos(half).unsqueeze(-1), axis * torch.sin(half).unsqueeze(-1)], dim=-1))


def quater_to_axis_theta(quater: torch.Tensor):
    """quaternion (…,4) -> (axis (…,3), angle (…) in [0, 2 pi]); the axis of a null rotation is the zero vector."""
    q = normalize(quater)
    cosa = q[..., 0]
    sina = torch.sqrt(1 - cosa ** 2).unsqueeze(-1)
    axis = q[..., 1:] / torch.max(sina, (sina < 1e-8).float())
    return axis, 2 * torch.acos(torch.clamp(cosa, min=-1, max=1))


def axis_theta_to_matrix(axis: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    return unit_quaternion_to_matrix(axis_theta_to_quater(axis, theta))


def matrix_to_axis_theta(matrix: torch.Tensor):
    return quater_to_axis_theta(matrix_to_unit_quaternion(matrix))


def matrix_to_rotvec(matrix: torch.Tensor) -> torch.Tensor:
    """Rotation vector with the reference's angle convention: theta mod 2 pi, plus 2 pi."""
    import math
    axis, theta = matrix_to_axis_theta(matrix)
    return axis * (theta % (2 * math.pi) + 2 * math.pi).unsqueeze(-1)


def rotvec_to_axis_theta(rotvec: torch.Tensor):
    theta = torch.norm(rotvec, dim=-1, keepdim=True)
    return rotvec / torch.max(theta, (theta < 1e-8).float()), theta.squeeze(-1)


def rotvec_to_matrix(rotvec: torch.Tensor) -> torch.Tensor:
    return axis_theta_to_matrix(*rotvec_to_axis_theta(rotvec))


def rot_diff_rad(rot1, rot2, yaxis_only=False):      # also exported here, where the reference keeps them (rotations.py:345+)
    from .metrics import rot_diff_rad as _impl
    return _impl(rot1, rot2, yaxis_only=yaxis_only)


def rot_diff_degree(rot1, rot2, yaxis_only=False):
    from .metrics import rot_diff_degree as _impl
    return _impl(rot1, rot2, yaxis_only=yaxis_only)
