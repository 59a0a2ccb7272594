// One selection round of the register-resident furthest-point sampler (BLOCKED ownership, 16 points per lane, four waves), as
// device functions shared by fps_kernel_blocked (csrc/fps.hip) and the level-1 stream kernel (csrc/sa_bf16.hip), whose sampler
// workgroups run the same loop and publish their picks while the rest of the chip consumes them.
// Contract (reference sampling_gpu.cu:93-140 / SURVEY.md §8 a1): distance ((dx*dx+dy*dy)+dz*dz) unfused fp32, running minimum,
// strict '>' arg max = lowest index among equal maxima.
#pragma once
#include "common.h"

typedef float fps_f32x2 __attribute__((ext_vector_type(2)));

// u32 max reductions whose DPP move folds into the max (v_max_u32_dpp): 0 is the identity, so lanes without a valid
// source (bound_ctrl) or in masked-off rows simply contribute 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    return max(v, o);
}
__device__ __forceinline__ unsigned row_max_u32_fold(unsigned v) {
    v = dpp_max_u32<0xB1, 0xF>(v);
    v = dpp_max_u32<0x4E, 0xF>(v);
    v = dpp_max_u32<0x141, 0xF>(v);
    v = dpp_max_u32<0x140, 0xF>(v);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32_fold(unsigned v) {
    v = row_max_u32_fold(v);
    v = dpp_max_u32<0x142, 0xA>(v);  // row_bcast15 -> rows 1,3
    v = dpp_max_u32<0x143, 0xC>(v);  // row_bcast31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// A lane's sixteen points against the last pick (ox, oy, oz): running minima updated (on the bit patterns: distances are >= 0),
// `best` = the largest of them, `li` = its lowest slot -- the maximum through a tree of pairwise maxima, the slot as a descent
// through that tree (at every level the LEFT half wins when it holds `best`: lowest slot among equals).
__device__ __forceinline__ void fps_lane_round16(const fps_f32x2 (&px)[8], const fps_f32x2 (&py)[8], const fps_f32x2 (&pz)[8],
                                                 unsigned (&dmin)[16], float ox, float oy, float oz, unsigned &best, int &li) {
    const fps_f32x2 o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz};
    unsigned q8[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const fps_f32x2 dx = px[h] - o2x, dy = py[h] - o2y, dz = pz[h] - o2z;
        const fps_f32x2 d = (dx * dx + dy * dy) + dz * dz;
        const unsigned b0 = __float_as_uint(d[0]), b1 = __float_as_uint(d[1]);
        dmin[2 * h] = b0 < dmin[2 * h] ? b0 : dmin[2 * h];
        dmin[2 * h + 1] = b1 < dmin[2 * h + 1] ? b1 : dmin[2 * h + 1];
        q8[h] = max(dmin[2 * h], dmin[2 * h + 1]);
    }
    unsigned q4[4], q2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) q4[i] = max(q8[2 * i], q8[2 * i + 1]);
    q2[0] = max(q4[0], q4[1]); q2[1] = max(q4[2], q4[3]);
    best = max(q2[0], q2[1]);
    const bool h3 = q2[0] != best;                                   // the maximum is in slots 8..15 only
    const unsigned a4 = h3 ? q4[2] : q4[0];
    const bool h2 = a4 != best;
    const unsigned a8l = h3 ? (h2 ? q8[6] : q8[4]) : (h2 ? q8[2] : q8[0]);
    const bool h1 = a8l != best;
    const int p = (h3 ? 4 : 0) + (h2 ? 2 : 0) + (h1 ? 1 : 0);          // pair index 0..7
    unsigned dl = dmin[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) dl = p == i ? dmin[2 * i] : dl;
    li = 2 * p + (dl != best ? 1 : 0);
}

// The wave's winner: (largest minimum, index of its lowest holder) -- one DPP max reduction, then ballot + find-first-set +
// v_readlane (blocked ownership: lowest lane = lowest indices).  `base` = the lane's first point.
__device__ __forceinline__ void fps_wave_winner(unsigned best, int li, int base, unsigned &wmax, unsigned &widx) {
    wmax = wave_max_u32_fold(best);
    const unsigned long long hit = __ballot(best == wmax);
    const int wl = __ffsll((long long)hit) - 1;
    widx = (unsigned)__builtin_amdgcn_readlane(base + li, wl);
}

// Four waves: every lane reads the four (maximum, index) pairs as two 16-byte broadcasts and picks the winner with three strict
// compares in wave order (ties: the lower wave = the lower indices) -- no cross-lane reduction, ballot or readlane on the round's
// critical chain; the pick stays in a (uniform) vector register.
__device__ __forceinline__ int fps_winner_of_four(const uint2 *slot) {
    const uint4 s01 = *reinterpret_cast<const uint4 *>(slot), s23 = *reinterpret_cast<const uint4 *>(slot + 2);
    unsigned bv = s01.x, bi = s01.y;
    bi = s01.z > bv ? s01.w : bi; bv = s01.z > bv ? s01.z : bv;
    bi = s23.x > bv ? s23.y : bi; bv = s23.x > bv ? s23.x : bv;
    bi = s23.z > bv ? s23.w : bi;
    return (int)bi;
}

// The same for a lane holding PPT (even, != 16) points: maximum by a linear walk, lowest slot by a descending select chain.
template <int PPT>
__device__ __forceinline__ void fps_lane_round(const fps_f32x2 (&px)[PPT / 2], const fps_f32x2 (&py)[PPT / 2], const fps_f32x2 (&pz)[PPT / 2],
                                               unsigned (&dmin)[PPT], float ox, float oy, float oz, unsigned &best, int &li) {
    const fps_f32x2 o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz};
    best = 0u;
#pragma unroll
    for (int h = 0; h < PPT / 2; ++h) {
        const fps_f32x2 dx = px[h] - o2x, dy = py[h] - o2y, dz = pz[h] - o2z;
        const fps_f32x2 d = (dx * dx + dy * dy) + dz * dz;
        const unsigned b0 = __float_as_uint(d[0]), b1 = __float_as_uint(d[1]);
        dmin[2 * h] = b0 < dmin[2 * h] ? b0 : dmin[2 * h];
        dmin[2 * h + 1] = b1 < dmin[2 * h + 1] ? b1 : dmin[2 * h + 1];
        best = max(best, max(dmin[2 * h], dmin[2 * h + 1]));
    }
    li = PPT - 1;
#pragma unroll
    for (int i = PPT - 2; i >= 0; --i) li = dmin[i] == best ? i : li;
}
