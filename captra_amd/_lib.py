"""ctypes binding of libcaptra_hip.so (the C ABI declared in include/captra_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a launch fails this
module raises.  Tensors are handed over as raw device pointers plus the current torch HIP stream,
which is what the reference's glue does with `at::cuda::getCurrentCUDAStream()`
(network/models/pointnet_lib/src/ball_query.cpp:22).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
import os
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["CAPTRA_LIB"]) if os.environ.get("CAPTRA_LIB") else _PKG / "lib" / "libcaptra_hip.so"   # (CAPTRA_LIB: same-box A/B of two builds)

_lib = None

_INT, _LL, _F, _P = C.c_int, C.c_longlong, C.c_float, C.c_void_p

# name -> argtypes (all return int unless listed in _VOID / _OTHER)
_SIGNATURES = {
    "captra_furthest_point_sampling": [_INT, _INT, _INT, _P, _P, _P, _P],
    "captra_ball_query": [_INT, _INT, _INT, _F, _INT, _P, _P, _P, _P],
    "captra_group_points": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P],
    "captra_group_points_multi": [_INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P],
    "captra_group_points_grad": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P],
    "captra_group_points_grad_ws": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, C.c_size_t, _P],
    "captra_three_interpolate_grad_ws": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, C.c_size_t, _P],
    "captra_gather_points": [_INT, _INT, _INT, _INT, _P, _P, _P, _P],
    "captra_gather_points_grad": [_INT, _INT, _INT, _INT, _P, _P, _P, _P],
    "captra_gather_points_grad_ws": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, C.c_size_t, _P],
    "captra_knn": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_three_nn": [_INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_three_interpolate": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_three_interpolate_grad": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_canonicalize": [_INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_canonicalize_planes": [_INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_ball_query_multi": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_pack_weights": [_INT, _INT, _P, _P, _P, _P, _P],
    "captra_pointwise_mlp": [_INT, _INT, _INT, _LL, _P, _P, _P, _INT, _P, _P],
    "captra_pointwise_mlp_cb": [_INT, _INT, _INT, _LL, _P, _P, _P, _INT, _P, _P],
    "captra_pointwise_mlp2": [_INT, _INT, _INT, _INT, _LL, _P, _P, _INT, _P, _P, _INT, _P, _P],
    "captra_sa_group_mlp": [_INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_mlp_max": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, _INT, _INT, _P],
    "captra_sa_scale_fused": [_INT] * 8 + [_P] * 11 + [_INT, _INT, _P],
    "captra_three_nn_weights": [_INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_pointwise_mlp_gn": [_INT, _INT, _INT, _LL, _P, _P, _P, _P, _INT, _P, _P, _INT, _P],
    "captra_gn_finalize": [_INT, _INT, _INT, _INT, _LL, _F, _P, _P, _P, _P, _P],
    "captra_gn_finalize_tm": [_INT, _INT, _INT, _INT, _LL, _F, _P, _P, _P, _P, _P],
    "captra_pack_weights_bf16": [_INT, _INT, _P, _P, _P],
    "captra_pack_weights_frag": [_INT, _INT, _P, _P],
    "captra_pointwise_mlp_bf16": [_INT, _INT, _INT, _LL, _P, _P, _P, _INT, _P, _P],
    "captra_pointwise_mlp_bf16_pm": [_INT, _INT, _INT, _LL, _P, _P, _P, _INT, _P, _P],
    "captra_pack_dense_bf16": [_INT, _INT, _INT, _P, _P, _P],
    "captra_pointwise_mlp_bf16pm": [_INT, _INT, _INT, _LL, _INT, _P, _P, _P, _P, _INT, _INT, _P, _P],
    "captra_pointwise_mlp_bf16pm_cb": [_INT, _INT, _INT, _LL, _INT, _P, _P, _P, _INT, _INT, _P, _P],
    "captra_pointwise_mlp_bf16pm_stats": [_INT, _INT, _INT, _LL, _INT, _P, _P, _P, _P, _INT, _P, _P, _P],
    "captra_gn_stats_bf16pm": [_INT, _INT, _LL, _P, _P, _P],
    "captra_dense_bf16_tile": [_INT, _INT, _INT, _LL, _P, _P, _P, _LL, _P, _INT, _P, _P, _P],
    "captra_dense_bf16_tile_ex": [_INT, _INT, _INT, _LL, _INT, _P, _P, _INT, _P, _P, _LL, _P, _INT, _INT, _P, _P, _P],
    "captra_gemv_bf16": [_INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_head12_bf16": [_INT, _INT, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_mlp_chain_bf16": [_INT, _INT, _LL, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_pack_sa_bf16": [_INT] * 5 + [_P] * 7 + [_P],
    "captra_sa_scale_bf16": [_INT] * 9 + [_P] * 6 + [_INT, _INT, _P],
    "captra_pack_sa_x6": [_INT] * 4 + [_P] * 6 + [_P],
    "captra_sa_scales_multi": [_INT, _P, _P, _P],
    "captra_pack_chain_x6": [_INT, _INT, _INT, _INT, _LL, _LL, _P, _P, _P, _P],
    "captra_mlp_chain3_x6": [_INT, _INT, _LL, _P, _P, _INT, _P, _P],
    "captra_coord_tail_x6": [_INT, _INT, _INT, _INT, _LL, _P, _P, _INT, _P, _P, _P],
    "captra_pack_dense_x6": [_INT, _INT, _P, _P, _P],
    "captra_pointwise_mlp_x6": [_INT, _INT, _INT, _LL, _P, _P, _P, _P, _INT, _P, _P, _INT, _P],
    "captra_sa_scale_x6": [_INT] * 9 + [_P] * 7 + [_INT, _INT, _P],
    "captra_query_and_group": [_INT, _INT, _INT, _F, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_neck_chain_bf16": [_INT, _INT, _LL, _INT, _P, _P, _P, _INT, _P, _P, _P, _P, _INT, _P, _P, _INT, _INT, _P, _P],
    "captra_bq_planes": [_INT, _INT, _P, _P, _P],
    "captra_sa1_stream_bf16": [_INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _INT, _P, _P, _P, _INT, _P, _P, _P, _INT, _P, _P, _P, _P, _P],
    "captra_coord_tail": [_INT, _INT, _INT, _INT, _LL, _P, _P, _P, _INT, _P, _P, _P],
    "captra_fps_gather": [_INT, _INT, _INT, _P, _P, _P, _P, _P],
    "captra_fps_gather_ragged": [_INT, _INT, _P, _INT, _P, _P, _P, _P, _P],
    "captra_fps_gather_part": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_crop_ball": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_crop_box": [_INT, _INT, _INT, C.c_double, _P, _P, _P, _P, _P, _P, _P],
    "captra_otf_candidates": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_otf_finish": [_INT, _INT, _INT, _INT] + [_P] * 11 + [_P],
    "captra_sa_scale_pre": [_INT] * 8 + [_P] * 9 + [_P, _INT, _INT, _P],
    "captra_sa_scale_pre_pm": [_INT] * 8 + [_P] * 9 + [_P, _INT, _INT, _P],
    "captra_pointwise_mlp_pm": [_INT, _INT, _INT, _LL, _P, _P, _P, _INT, _P, _P],
    "captra_rot_pool_compose": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_mlp_chain3": [_INT, _INT, _INT, _INT, _INT, _LL, _P, _P, _P, _P, _P, _P, _P, _INT, _P, _P],
    "captra_interp_concat": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P],
    "captra_fp_interpolate_concat": [_INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_group_norm_relu": [_INT, _INT, _INT, _INT, _F, _INT, _P, _P, _P, _P, _P],
    "captra_part_fit_st": [_INT, _INT, _INT, _INT, _P, _P, _P, _INT, _P, _P, _P, _P, _P, _P],
    "captra_part_fit_st_track": [_INT, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "captra_seg_softmax_argmax": [_INT, _INT, _INT, _P, _P, _P, _P],
    "captra_copy_multi": [_INT, _P, _P, _P, _P],
    "captra_row_max": [_LL, _INT, _P, _P, _P],
    "captra_pack_pose": [_INT, _P, _P, _P, _P, _P, _P, _P],
    "captra_procrustes_rot3": [_INT, _INT, _P, _P, _P, _P],
}


class CaptraHipError(RuntimeError):
    pass


# ---- per-call launch options (include/captra_hip.h captra_launch_opts) ---------------------------------------------------------
# The C library holds no product-affecting state: what a caller wants different from the defaults travels with every call.  The
# HOST layer keeps the calling thread's wishes here (fused.split_k / fused.centre_window / graph.BackbonePipe set them) and `call`
# hands them to the `_ex` form of the entry points that read them.
class LaunchOpts(C.Structure):
    _fields_ = [("splitk_positions", _INT), ("sa_prezeroed", _INT), ("centre_m0", _INT), ("centre_mc", _INT), ("reserved_cus", _INT),
                ("dyn_slot", _P)]


class SaScaleJob(C.Structure):
    _fields_ = ([(k, _INT) for k in ("pre", "b", "n", "m", "k", "cfeat", "c1", "c2", "c3")]
                + [(k, _P) for k in ("feat_or_v1", "xyz_cn", "new_xyz", "idx", "w1", "b1", "w2", "b2", "w3", "b3", "out")]
                + [("out_ctotal", _INT), ("co_off", _INT)])


EX_ENTRY_POINTS = {"captra_ball_query", "captra_ball_query_multi", "captra_pointwise_mlp", "captra_pointwise_mlp2", "captra_pointwise_mlp_pm",
                   "captra_pointwise_mlp_gn", "captra_sa_scale_fused", "captra_sa_scale_pre_pm", "captra_sa_scale_bf16", "captra_head12_bf16"}
_OPTS_TLS = threading.local()


class _OptState:
    __slots__ = ("splitk_positions", "sa_prezeroed", "centre_m0", "centre_mc", "reserved_cus", "dyn_pool", "dyn_slots", "dyn_next")

    def __init__(self):
        self.splitk_positions = self.sa_prezeroed = self.centre_m0 = self.centre_mc = self.reserved_cus = 0
        self.dyn_pool, self.dyn_slots, self.dyn_next = 0, 0, 0


def opt_state() -> _OptState:
    st = getattr(_OPTS_TLS, "st", None)
    if st is None:
        st = _OPTS_TLS.st = _OptState()
    return st


@contextlib.contextmanager
def launch_options(**kw):
    """Options of the calling thread's launches inside the block (fields of captra_launch_opts; `dyn_pool` = (device pointer of
    an int32 buffer, slots): every launch that hands its centres out dynamically takes the next slot, round robin).  Restored on exit."""
    st = opt_state()
    prev = {k: getattr(st, k) for k in _OptState.__slots__}
    for k, v in kw.items():
        if k == "dyn_pool":
            st.dyn_pool, st.dyn_slots, st.dyn_next = (int(v[0]), int(v[1]), 0) if v else (0, 0, 0)
        else:
            setattr(st, k, int(v))
    try:
        yield
    finally:
        for k, v in prev.items():
            setattr(st, k, v)


def current_opts(name: str | None = None):
    """The calling thread's options as a captra_launch_opts (None when they are the defaults)."""
    st = opt_state()
    dyn = 0
    if st.dyn_slots > 0 and name in ("captra_sa_scale_fused", "captra_sa_scale_pre_pm"):
        dyn = st.dyn_pool + 4 * (st.dyn_next % st.dyn_slots)
        st.dyn_next += 1
    if not (st.splitk_positions or st.sa_prezeroed or st.centre_mc > 0 or st.reserved_cus or dyn):
        return None
    return LaunchOpts(st.splitk_positions, st.sa_prezeroed, st.centre_m0, st.centre_mc, st.reserved_cus, dyn or None)


def lib():
    """Load libcaptra_hip.so once; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise CaptraHipError(
                f"{LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `python captra_amd/build.py`). captra_amd has no CPU fallback.")
        l = C.CDLL(str(LIB_PATH))
        for name, args in _SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                continue  # a stale build: call() raises when the symbol is actually needed
            fn.argtypes = args
            fn.restype = _INT
            if name in EX_ENTRY_POINTS:
                fx = getattr(l, name + "_ex")
                fx.argtypes = args[:-1] + [_P, _P]
                fx.restype = _INT
        l.captra_error_string.argtypes = [_INT]
        l.captra_error_string.restype = C.c_char_p
        l.captra_version.restype = C.c_char_p
        l.captra_prof_enable.argtypes = [_INT]
        l.captra_prof_enable.restype = None
        l.captra_prof_reset.restype = None
        l.captra_prof_read.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        l.captra_prof_read.restype = _INT
        l.captra_prof_names.argtypes = [C.c_char_p, _INT]
        l.captra_prof_names.restype = _INT
        for name, args in (("captra_group_points_grad_ws_bytes", [_INT] * 5), ("captra_three_interpolate_grad_ws_bytes", [_INT] * 4),
                           ("captra_gather_points_grad_ws_bytes", [_INT] * 4)):
            if hasattr(l, name):
                getattr(l, name).argtypes = args
                getattr(l, name).restype = C.c_size_t
        if hasattr(l, "captra_packed_weight_floats"):
            l.captra_packed_weight_floats.argtypes = [_INT, _INT]
            l.captra_packed_weight_floats.restype = _LL
        if hasattr(l, "captra_dense_bf16_image_bytes"):
            l.captra_dense_bf16_image_bytes.argtypes = [_INT, _INT]
            l.captra_dense_bf16_image_bytes.restype = _LL
            l.captra_gn_stats_bf16pm_tiles.argtypes = [_LL]
            l.captra_gn_stats_bf16pm_tiles.restype = _INT
            l.captra_dense_bf16_stats_tiles.argtypes = [_LL]
            l.captra_dense_bf16_stats_tiles.restype = _INT
        if hasattr(l, "captra_dense_bf16_tile_stats_tiles"):
            l.captra_dense_bf16_tile_stats_tiles.argtypes = [_LL]
            l.captra_dense_bf16_tile_stats_tiles.restype = _INT
        if hasattr(l, "captra_chain_bf16_image_bytes"):
            l.captra_chain_bf16_image_bytes.argtypes = [_INT, _INT]
            l.captra_chain_bf16_image_bytes.restype = _LL
        if hasattr(l, "captra_dense_x6_image_bytes"):
            l.captra_dense_x6_image_bytes.argtypes = [_INT, _INT]
            l.captra_dense_x6_image_bytes.restype = _LL
        if hasattr(l, "captra_sa_x6_image_bytes"):
            l.captra_sa_x6_image_bytes.argtypes = [_INT] * 4
            l.captra_sa_x6_image_bytes.restype = _LL
        if hasattr(l, "captra_sa_bf16_image_bytes"):
            l.captra_sa_bf16_image_bytes.argtypes = [_INT] * 4
            l.captra_sa_bf16_image_bytes.restype = _LL
        if hasattr(l, "captra_sa1_stream_scratch_bytes"):
            l.captra_sa1_stream_scratch_bytes.argtypes = [_INT, _INT]
            l.captra_sa1_stream_scratch_bytes.restype = _LL
            l.captra_sa1_stream_set_grid.argtypes = [_INT, _INT]
            l.captra_sa1_stream_set_grid.restype = None
            l.captra_sa1_stream_set_fine.argtypes = [_INT]
            l.captra_sa1_stream_set_fine.restype = None
            l.captra_sa1_stream_set_whole.argtypes = [_INT]
            l.captra_sa1_stream_set_whole.restype = None
        if hasattr(l, "captra_neck_chain_set_split"):
            l.captra_neck_chain_set_split.argtypes = [_INT]
            l.captra_neck_chain_set_split.restype = None
        if hasattr(l, "captra_query_and_group_set_shape"):
            l.captra_query_and_group_set_shape.argtypes = [_INT, _INT]
            l.captra_query_and_group_set_shape.restype = None
        if hasattr(l, "captra_pointwise_mlp_gn_tiles"):
            l.captra_pointwise_mlp_gn_tiles.argtypes = [_INT, _INT, _LL]
            l.captra_pointwise_mlp_gn_tiles.restype = _INT
            l.captra_pointwise_mlp_gn_tiles_ex.argtypes = [_INT, _INT, _LL, _P]
            l.captra_pointwise_mlp_gn_tiles_ex.restype = _INT
        _lib = l
        # A/B switches for measurements (thread-local knobs of the library, set for the importing thread)
        import os
        for env, fn in (("CAPTRA_BF16_SHARED_AFFINE", "captra_dense_bf16_set_shared_affine"), ("CAPTRA_SA_BF16_VARIANT", "captra_sa_bf16_set_variant"),
                        ("CAPTRA_FPS_DEFER", "captra_fps_set_defer"), ("CAPTRA_NN_SPLIT", "captra_three_nn_set_split"),
                        ("CAPTRA_BQ_CPW", "captra_ball_query_set_cpw"), ("CAPTRA_SA_SPLIT", "captra_sa_fused_set_split"),
                        ("CAPTRA_HEAD_PERSIST", "captra_tile_bf16_set_persistent")):
            if env in os.environ and hasattr(l, fn):
                getattr(l, fn)(C.c_int(int(os.environ[env])))
        if "CAPTRA_L1_FINE" in os.environ and hasattr(l, "captra_sa1_stream_set_fine"):
            l.captra_sa1_stream_set_fine(int(os.environ["CAPTRA_L1_FINE"]))
        if "CAPTRA_L1_WHOLE" in os.environ and hasattr(l, "captra_sa1_stream_set_whole"):
            l.captra_sa1_stream_set_whole(int(os.environ["CAPTRA_L1_WHOLE"]))
        if "CAPTRA_L1_GRID" in os.environ and hasattr(l, "captra_sa1_stream_set_grid"):
            l.captra_sa1_stream_set_grid(int(os.environ["CAPTRA_L1_GRID"]), 1)
    return _lib


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def check(err: int, what: str) -> None:
    if err != 0:
        msg = lib().captra_error_string(err)
        raise CaptraHipError(f"{what} failed: {msg.decode() if msg else err} (code {err})")


def require_device(*tensors) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise CaptraHipError("captra_amd ops need tensors in GPU memory (got a CPU tensor); "
                                 "there is no CPU fallback in the product path")
        if not t.is_contiguous():
            raise CaptraHipError("captra_amd ops need contiguous tensors")


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def call(name: str, *args) -> None:
    """Invoke a launcher with the current stream appended and raise on a non-zero return."""
    try:
        fn = getattr(lib(), name)
    except AttributeError as e:
        raise CaptraHipError(f"{LIB_PATH} does not export {name}: rebuild it (python captra_amd/build.py)") from e
    if name in EX_ENTRY_POINTS:
        o = current_opts(name)
        if o is not None:
            check(getattr(lib(), name + "_ex")(*args, C.byref(o), stream_ptr()), name + "_ex")
            return
    check(fn(*args, stream_ptr()), name)


# ---- profiling helpers ---------------------------------------------------------------------------
def prof_enable(on: bool) -> None:
    lib().captra_prof_enable(1 if on else 0)


def prof_reset() -> None:
    lib().captra_prof_reset()


def prof_read(name: str) -> tuple[float, int]:
    ms, n = C.c_double(0.0), C.c_longlong(0)
    lib().captra_prof_read(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


def prof_names() -> list[str]:
    buf = C.create_string_buffer(4096)
    lib().captra_prof_names(buf, 4096)
    s = buf.value.decode()
    return [x for x in s.split(",") if x]
