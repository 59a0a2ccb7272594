// Shared device/host helpers for libcaptra_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "captra_hip.h"   // captra_launch_opts, captra_stream_t

#define CAPTRA_WAVE 64


// ---- profiling hooks (prof.cpp) -------------------------------------------------------------
// CAPTRA_LAUNCH brackets a kernel launch with HIP events on the launch stream when profiling
// is enabled (captra_prof_enable).  Outside profiling it costs one relaxed load.
struct CaptraProfScope {
    int slot;
    hipStream_t stream;
    CaptraProfScope(const char *name, hipStream_t s);
    ~CaptraProfScope();
};

#define CAPTRA_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                     \
    do {                                                                                 \
        CaptraProfScope _scope(name, stream);                                            \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);             \
    } while (0)

// In-kernel phase timers of the SA kernels (captra_sa_fused_set_prof, tools/bench_sa_fused.py --phases) exist only in a build with
// -DCAPTRA_SA_PROF=1 (CAPTRA_HIPCC_EXTRA): the runtime test `p.prof != nullptr` alone -- a branch at every phase boundary, which
// the scheduler cannot move anything across -- costs the fp32 SA1 scales 2.7 %, the SA2 scales 1-2 % and cost the bf16 SA2 kernel
// a third (same-box A/B of two builds, CAPTRA_LIB).
#ifndef CAPTRA_SA_PROF
#define CAPTRA_SA_PROF 0
#endif
#define CAPTRA_PROF_ON(ptr) (CAPTRA_SA_PROF != 0 && (ptr) != nullptr)

// Ablation instantiations and timing modes whose RESULTS ARE WRONG BY CONSTRUCTION (measurement only: tools/bench_l1_stream.py,
// tools/bf16_variants.sh) exist only in a build with -DCAPTRA_ABLATIONS=1 (CAPTRA_HIPCC_EXTRA); the shipped library ignores their knobs.
#ifndef CAPTRA_ABLATIONS
#define CAPTRA_ABLATIONS 0
#endif

// Per-call options (include/captra_hip.h captra_launch_opts; NULL = defaults): the library keeps no product-affecting state.
// CUs a persistent launch leaves free
static inline int captra_reserved_cus(const captra_launch_opts *o) { return (o != nullptr && o->reserved_cus > 0) ? o->reserved_cus : 0; }
// the call's centre window clipped to [0, m); true when one is given
static inline bool captra_centre_window(const captra_launch_opts *o, int m, int *m0, int *mc) {
    if (o == nullptr || o->centre_mc <= 0) { *m0 = 0; *mc = m; return false; }
    const int w0 = o->centre_m0 < 0 ? 0 : o->centre_m0;
    *m0 = w0 < m ? w0 : m;
    *mc = o->centre_mc < m - *m0 ? o->centre_mc : m - *m0;
    return true;
}

static inline int captra_last_error() { return (int)hipGetLastError(); }

// ---- per-device one-shot (kernel function attributes) ---------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: a process that launches on a second GPU must set it
// there too.  Usage:  if (once.first_use()) { hipFuncSetAttribute(...); once.done(); }
// first_use() only LOOKS (true until done() ran for the current device): the device is marked after the attribute calls
// returned, so a second host thread racing the first either sees the mark (attributes already applied) or applies them
// itself (idempotent) -- it can never launch with more dynamic LDS than the attribute allows yet.  A caller that never
// calls done() simply re-applies the attribute on every launch (correct, a few microseconds).
#include <atomic>
struct CaptraDeviceOnce {
    std::atomic<unsigned long long> seen[2] = {{0ull}, {0ull}};   // device ordinals 0..127
    static int device() {
        int dev = 0;
        return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 128) ? dev : -1;
    }
    bool first_use() const {
        const int dev = device();
        if (dev < 0) return true;                                  // unknown device: always (re)set
        return (seen[dev >> 6].load(std::memory_order_acquire) & (1ull << (dev & 63))) == 0ull;
    }
    void done() {
        const int dev = device();
        if (dev >= 0) seen[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
};

// debug / experiment knobs (captra_*_set_*): thread-local so that one host thread's A/B switch never changes what
// another thread (another GPU's stream in the same process) launches; the reference boundary has no global state.
#define CAPTRA_KNOB thread_local

// Zero `nbytes` (a multiple of 16, 16-byte aligned) on `stream` with a KERNEL (prof.cpp).  Not hipMemsetAsync: inside a captured
// hipGraph a memset node is not a kernel node, and in a graph that is ONE linear chain (the step with the networks one after the
// other) the fill was observed to overlap the kernel behind it at 64+ KiB -- the level-1 stream kernel's granules zeroed after its
// samplers had published them (consumers gave up waiting, garbage picks); a kernel node is ordered like every other launch.
int captra_zero_async(void *p, size_t nbytes, hipStream_t stream);

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// squared distance exactly as the reference kernels and the oracle write it:
// ((dx*dx + dy*dy) + dz*dz), every operation rounded separately (build uses -ffp-contract=off).
__device__ __forceinline__ float dist2_unfused(float ax, float ay, float az, float bx, float by,
                                               float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// 64-bit shuffle helpers (wave64)
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = shfl_xor_u64(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// activation codes of the shared-MLP entry points (include/captra_hip.h: CAPTRA_ACT_*)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID_M05 = 2 };

// ReLU on the bit pattern: negative floats (and -0) are negative integers, so one v_max_i32 does it; written on floats the
// compiler emits a canonicalising v_max_f32 x, x, x in front of the v_max_f32 x, 0 (IEEE mode).  NaNs with the sign bit
// clear pass through, as in the reference's relu.
__device__ __forceinline__ float relu_bits(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(x > 0 ? x : 0);
}
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return relu_bits(v);
    if (act == ACT_SIGMOID_M05) return 1.0f / (1.0f + expf(-v)) - 0.5f;
    return v;
}
