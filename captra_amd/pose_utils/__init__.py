"""Pose algebra of the tracking loop (mirrors the reference's pose_utils/ for the functions the
hot path calls — SURVEY.md §2 rows 12-13)."""
from . import metrics, part_dof_utils, pose_fit, procrustes, rotations  # noqa: F401
