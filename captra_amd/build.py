"""Build recipe for libcaptra_hip.so (gfx950 only) and the CPU oracle.

hipcc cross-compiles without a GPU, so this runs in the authoring container as well as on the
GPU box.  Objects are cached next to the sources (csrc/_obj/) and rebuilt when a source or a
header is newer.  The shared library is written IN-TREE (captra_amd/lib/) so that it travels with
the repository snapshot to the GPU box.

Replaces the reference's `CUDAExtension('pointnet2_cuda', ..., nvcc -O2)` recipe
(network/models/pointnet_lib/setup.py:4-23).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libcaptra_hip.so"
ARCH = "gfx950"

# -ffp-contract=off: the bit-exact contract of FPS / ball query / three_nn is written in terms of
# separately rounded fp32 operations; hipcc contracts to FMA by default.
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    # sa_fused.hip's register-resident kernels are fully unrolled by design (activations are statically indexed
    # registers): lift LLVM's cap on "#pragma unroll" bodies, else the widest shape falls back to scratch arrays
    "-mllvm", "-pragma-unroll-threshold=200000",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-result",
    "-Wno-unused-value",
    f"-I{ROOT / 'include'}",
    f"-I{CSRC}",
    *os.environ.get("CAPTRA_HIPCC_EXTRA", "").split(),   # experiments only (e.g. "-mllvm -amdgpu-sched-strategy=max-ilp")
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _sources() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _headers() -> list[Path]:
    return sorted(list(CSRC.glob("*.h")) + list((ROOT / "include").glob("*.h")))


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


# Per-source scheduling strategy.  LLVM's default GCN strategy schedules for occupancy first; the register-resident SA
# kernels, the FP1 chain / CoordNet tail and the plain FPS kernel fix their occupancy by construction and gain from the
# latency-first one (same-box A/B on MI355X: SA family 3.887 -> 3.846 ms per step, coord_tail 0.184 -> 0.175, FPS 0.349 ->
# 0.327), the dense-layer kernels (pointwise_mlp.hip: 1.636 -> 1.657) and the pruned sampler (+4 %) lose and stay on the default.
# bf16_mlp.hip loses 5 % with it, interpolate.hip / ball_query.hip do not move.  Instruction order only: every result is bit-identical.
MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# sa_bf16.hip (round 5): the level-1 stream kernel's sampler rounds are 173 instructions with it and 200 (31 s_nop) without: 250 vs 274 us
# for the 511 rounds; the bf16 SA scales in the same file do not move (same-box A/B of two builds)
# sa_x6.hip / dense_x6.hip: no SLP vectorisation -- packed fp32 VALU (v_pk_add_f32) beside MFMAs costs more than the two scalar
# instructions it replaces (MI355X_MICROARCH.md "price of one filler beside MFMAs"); the operand split is written on scalars on purpose
NO_SLP = ["-fno-slp-vectorize"]
PER_SOURCE_FLAGS = {"sa_fused.hip": MAX_ILP, "mlp_chain.hip": MAX_ILP, "fps.hip": MAX_ILP, "sa_bf16.hip": MAX_ILP, "sa_x6.hip": NO_SLP,
                    "dense_x6.hip": NO_SLP, "chain_x6.hip": NO_SLP}


def _compile_one(src: Path, force: bool, verbose: bool) -> Path:
    obj = OBJ / (src.name + ".o")
    if force or _stale(obj, [src, Path(__file__)] + _headers()):
        cmd = [_hipcc(), *HIPCC_FLAGS, *PER_SOURCE_FLAGS.get(src.name, []), "-x", "hip", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
    return obj


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 and link libcaptra_hip.so. Returns the library path."""
    OBJ.mkdir(parents=True, exist_ok=True)
    LIBDIR.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, force, verbose), srcs))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB


def build_oracle(force: bool = False, verbose: bool = False) -> Path:
    """Compile the plain-C restatement (oracle/) with gcc; test infrastructure only."""
    odir = ROOT / "oracle"
    cmd = ["make", "-C", str(odir)] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(res.stdout)
    return odir / "libcaptra_oracle.so"


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force=force, verbose=True))
    if (ROOT / "oracle" / "Makefile").exists():
        print(build_oracle(force=force, verbose=True))
