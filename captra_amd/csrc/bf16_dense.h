// Shared device helpers of the bf16-native dense kernels (csrc/dense_bf16.hip, csrc/tile_bf16.hip): packing, the MFMA wrapper,
// the slot-order permutation, the statistics butterfly and the on-load GroupNorm transform.
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned db_pack(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ unsigned db_relu2(unsigned v) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
__device__ __forceinline__ f32x16 db_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__host__ __device__ __forceinline__ int db_perm(int s) { return (s & 3) | ((s & 4) << 1) | ((s & 8) >> 1); }

// ST epilogue: per-channel (sum, sum of squares) of what a wave just stored for one row tile -- 16 channels x (TN x 32)
// positions per half-wave -- reduced over the 32 columns by a halving butterfly (16 + 8 + 4 + 2 + 1 exchanges instead of
// 5 x 32: the lane with column c ends up holding value c of {s[0..15], q[0..15]}), fixed order, no atomics; one float per
// lane goes to stats[b][chunk][channel][0 / 1] (tile-major: a channel-major table would take 64 scattered 4-byte stores per
// wave and row tile, into 128-byte lines shared by 16 workgroups -- measured 4x the layer's own time).  The separate statistics pass (gn_stats_bf16pm_kernel) re-read the whole
// tensor: 134 MB and ~40 us per 512-wide layer at 32 x 4096 points.
template <int CNT>
__device__ __forceinline__ void db_stats_fold(float (&vals)[32], int col) {   // (every index a compile-time constant: registers)
    const bool up = (col & CNT) != 0;
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const float keep = up ? vals[k + CNT] : vals[k], send = up ? vals[k] : vals[k + CNT];
        vals[k] = keep + __shfl_xor(send, CNT, 64);
    }
}
__device__ __forceinline__ void db_stats_tile(float (&vals)[32], int col, int h, int row_tile, int cout, float *dst_b, int T, int chunk) {
    db_stats_fold<16>(vals, col);
    db_stats_fold<8>(vals, col);
    db_stats_fold<4>(vals, col);
    db_stats_fold<2>(vals, col);
    db_stats_fold<1>(vals, col);
    const int r = col & 15;
    const int ch = 32 * row_tile + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (ch < cout) dst_b[((size_t)chunk * cout + ch) * 2 + (col >> 4)] = vals[0];   // tile-major: a wave's 64 floats are 256 contiguous bytes
}
// accumulate the 8 stored values of one 16-byte store (acc registers 8 jj .. 8 jj + 7) into s / q
__device__ __forceinline__ void db_stats_acc(float (&vals)[32], int jj, u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(v[i] << 16), hi = __uint_as_float(v[i] & 0xffff0000u);
        vals[8 * jj + 2 * i] += lo;
        vals[8 * jj + 2 * i + 1] += hi;
        vals[16 + 8 * jj + 2 * i] = __builtin_fmaf(lo, lo, vals[16 + 8 * jj + 2 * i]);
        vals[16 + 8 * jj + 2 * i + 1] = __builtin_fmaf(hi, hi, vals[16 + 8 * jj + 2 * i + 1]);
    }
}

// x = bf16(relu(a * x + b)) on the 8 channels of a lane's B-operand registers; t = 16 floats (a0..a7, b0..b7) in LDS
__device__ __forceinline__ u32x4 db_affine(u32x4 v, const float *t) {
    const float4 a0 = *reinterpret_cast<const float4 *>(t), a1 = *reinterpret_cast<const float4 *>(t + 4);
    const float4 b0 = *reinterpret_cast<const float4 *>(t + 8), b1 = *reinterpret_cast<const float4 *>(t + 12);
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(v[i] << 16), hi = __uint_as_float(v[i] & 0xffff0000u);
        r[i] = db_relu2(db_pack(__builtin_fmaf(lo, a[2 * i], b[2 * i]), __builtin_fmaf(hi, a[2 * i + 1], b[2 * i + 1])));
    }
    return r;
}

}  // namespace
