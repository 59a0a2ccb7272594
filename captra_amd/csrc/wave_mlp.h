// Register-resident shared-MLP building blocks for gfx950 (wave64, v_mfma_f32_32x32x2_f32), shared by
// sa_fused.hip (set-abstraction scales) and mlp_chain.hip (dense layer chains).
//
// A wave owns 32 positions and carries them through consecutive 1x1-conv layers.  The MFMA accumulator layout
// (register r of lane l = row 8(r>>2)+(r&3)+4(l>>5), column l&31) becomes the next layer's B operand (lane-half h
// of k-step j = row 2j+h) with one v_permlane32_swap per register pair (sw_mid_epilogue), so activations never
// leave the vector registers.  Weights are streamed from the FRAGMENT image of the packed buffer (include/captra_hip.h
// "PACKED WEIGHTS": behind the row-major image; one 16-byte load per lane = four consecutive k-steps of an output tile)
// one 16-register set ahead of the MFMAs (sw_layer_reg).
// Accumulators start from the bias and K ascends within one wave: the k-ascending fmaf chain of the oracle.
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_maxf(float v) {
    v = fmaxf(v, dppf<0xB1>(v));
    v = fmaxf(v, dppf<0x4E>(v));
    v = fmaxf(v, dppf<0x141>(v));
    v = fmaxf(v, dppf<0x140>(v));
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppf_rm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

constexpr int pad32c(int c) { return (c + 31) / 32 * 32; }
constexpr int pad128c(int c) { return (c + 127) / 128 * 128; }

constexpr int SW_KS = 8;  // k-steps per register set and tile; a set = 2 tiles x 8 = 16 registers

template <int CIN, int COUT>
struct SwShape {
    static constexpr int KST = (CIN + 1) / 2;                 // MFMA k-steps (2 rows each)
    static constexpr int NSETS = (KST + SW_KS - 1) / SW_KS;
    static constexpr int NT = (COUT + 31) / 32;               // 32-row output tiles
    static constexpr int NPASS = (NT + 1) / 2;                // two tiles (independent accumulators) per pass
    static constexpr int STEPS = NPASS * NSETS;
    static constexpr int LDW = pad128c(COUT), KP = pad32c(CIN);
};

// the fragment image of a packed layer: element ((t*KQ + q)*64 + lane)*4 + i = W'^T[2(4q+i) + (lane>>5)][32t + (lane&31)]
template <int CIN, int COUT>
struct SwFrag {
    static constexpr int KST = (CIN + 1) / 2, KQ = (KST + 3) / 4, NT = (COUT + 31) / 32;
    static constexpr int BYTES = NT * KQ * 1024;
    static constexpr int OFFSET = pad32c(CIN) * pad128c(COUT);     // floats of the row-major image in front of it
};

template <int CIN, int COUT>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sw_frag_rsrc(const float *wt_packed) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)(wt_packed + SwFrag<CIN, COUT>::OFFSET), 0, SwFrag<CIN, COUT>::BYTES, 0x00020000);
}

// loads of set `c` of pass `ps` into dst[tm*8 + j]: two 16-byte loads per tile (8 k-steps).
// (bit_cast of the WHOLE vector: hipcc 7.2 lowers `bit_cast<float>(v[i])` on this builtin's result to a single
// buffer_load_dword and leaves the other three elements undefined)
template <int CIN, int COUT>
__device__ __forceinline__ void sw_load_set(float (&dst)[16], const __amdgpu_buffer_rsrc_t rsrc, int voff, int ps, int c) {
    using S = SwShape<CIN, COUT>;
    using F = SwFrag<CIN, COUT>;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int qq = 0; qq < SW_KS / 4; ++qq) {
            const int q = c * (SW_KS / 4) + qq, t = 2 * ps + tm;
            if (q < F::KQ && t < S::NT) {
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (t * F::KQ + q) * 1024, 0));
                dst[tm * SW_KS + qq * 4 + 0] = v.x; dst[tm * SW_KS + qq * 4 + 1] = v.y;
                dst[tm * SW_KS + qq * 4 + 2] = v.z; dst[tm * SW_KS + qq * 4 + 3] = v.w;
            }
        }
}

template <int CIN, int COUT>
__device__ __forceinline__ void sw_first_set(float (&dst)[16], const float *wt, int lane) {
    sw_load_set<CIN, COUT>(dst, sw_frag_rsrc<CIN, COUT>(wt), lane * 16, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void sw_bias_init(f32x16 &acc, const float *bias_lds, int t, int lane) {
    const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + 32 * t + 4 * (lane >> 5));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = bp[2 * q];
        acc[4 * q + 0] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
    }
}

// ReLU, then turn output tile t (rows 32t..32t+31) into B operands hout[16t..16t+15] (k-step = row pair)
template <int NOUT>
__device__ __forceinline__ void sw_mid_epilogue(const f32x16 &acc, int t, float (&hout)[NOUT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // ReLU on the bit pattern (one v_max_i32: negative floats and -0 are negative integers; the float form costs a
        // canonicalising v_max_f32 x, x, x in front of every v_max_f32 x, 0 -- 88 instructions per SA1 tile)
        unsigned a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = __float_as_int(acc[4 * q + i]);
            a[i] = (unsigned)(x > 0 ? x : 0);
        }
        // registers (4q, 4q+1) hold rows (8q, 8q+1) in the lower half-wave and (8q+4, 8q+5) in the upper one
        const auto p01 = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);
        const auto p23 = __builtin_amdgcn_permlane32_swap(a[2], a[3], false, false);
        const int k0 = 16 * t + 4 * q;
        if (k0 + 0 < NOUT) hout[k0 + 0] = __uint_as_float(p01[0]);  // rows 8q,   8q+1
        if (k0 + 1 < NOUT) hout[k0 + 1] = __uint_as_float(p23[0]);  // rows 8q+2, 8q+3
        if (k0 + 2 < NOUT) hout[k0 + 2] = __uint_as_float(p01[1]);  // rows 8q+4, 8q+5
        if (k0 + 3 < NOUT) hout[k0 + 3] = __uint_as_float(p23[1]);  // rows 8q+6, 8q+7
    }
}

// ReLU + max over the wave's 32 positions of output tile t -> red[row][wave], as a transpose-reduce butterfly (the form
// sa_pipe.hip issues in deferred parts; non-negative floats order like their bit patterns, so the max runs on integers): neighbouring REGISTERS
// are merged while neighbouring LANES are reduced -- stage k pairs lane l with l ^ 2^k and registers (2i, 2i+1); a lane with
// bit k clear keeps register 2i and takes the partner's 2i, a lane with bit k set keeps 2i+1 -- so four stages take 16 + 8 +
// 4 + 2 max steps instead of 16 x 4, ONE register then holds in lane l the maximum over its row of 16 lanes of accumulator
// register (l & 15); a ds_swizzle (lane ^ 16) joins the two rows of a half-wave, one ReLU (max on the raw bit patterns: a
// positive input wins the signed-integer maximum as the largest float, otherwise the result is negative and ReLU gives 0 =
// ReLU-then-max) and one 32-lane ds_write_b32 finish the tile: ~55 instructions instead of ~112.  These kernels are bound
// by the instructions a SIMD issues (MFMA issue + VALU do not overlap within it: the phase timers move when an
// instruction count moves and not when a latency does), so that is ~5 % of an SA1 scale.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ int sw_dpp_sel(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK_MASK, false);
}
__device__ __forceinline__ int sw_imax(int a, int b) { return a > b ? a : b; }
// max(own, oth of the partner lane) for a permutation in which EVERY lane has a source (quad_perm, row_ror): with
// bound_ctrl and a dead `old` the compiler folds the DPP move into the max (one v_max_i32_dpp instead of
// v_mov + v_mov_dpp + v_max)
template <int CTRL>
__device__ __forceinline__ int sw_imax_dpp(int own, int oth) {
    const int o = __builtin_amdgcn_update_dpp(0, oth, CTRL, 0xF, 0xF, true);
    return o > own ? o : own;
}

template <int COUT, int RED_STRIDE = 4>
__device__ __forceinline__ void sw_last_epilogue_bfly(const f32x16 &acc, int t, float *red, int wave, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    int w[8], v[4], u[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int x0 = __float_as_int(acc[2 * i]), x1 = __float_as_int(acc[2 * i + 1]);
        const int own = b0 ? x1 : x0, oth = b0 ? x0 : x1;
        w[i] = sw_imax_dpp<0xB1>(own, oth);                             // quad_perm [1,0,3,2]: lane ^ 1
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int own = b1 ? w[2 * j + 1] : w[2 * j], oth = b1 ? w[2 * j] : w[2 * j + 1];
        v[j] = sw_imax_dpp<0x4E>(own, oth);                             // quad_perm [2,3,0,1]: lane ^ 2
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int own = b2 ? v[2 * m + 1] : v[2 * m], oth = b2 ? v[2 * m] : v[2 * m + 1];
        const int p = sw_dpp_sel<0x104, 0x5>(own, oth);                  // lane ^ 4: row_shl:4 into banks 0, 2 ...
        const int q = sw_dpp_sel<0x114, 0xA>(p, oth);                    // ... row_shr:4 into banks 1, 3
        u[m] = sw_imax(own, q);
    }
    const int own = b3 ? u[1] : u[0], oth = b3 ? u[0] : u[1];
    int z = sw_imax_dpp<0x128>(own, oth);                                // row_ror:8: lane ^ 8 within the row of 16
    z = sw_imax(z, __builtin_amdgcn_ds_swizzle(z, 0x401F));              // lane ^ 16: the other row of the half-wave
    z = z > 0 ? z : 0;                                                   // ReLU on the bit pattern
    // lane l (l & 16 == 0) holds accumulator register r = l & 15 = output row 32 t + 8 (r >> 2) + (r & 3) + 4 (l >> 5)
    const int r = lane & 15;
    const int row = 32 * t + 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5);
    if ((lane & 16) == 0 && row < COUT) red[row * RED_STRIDE + wave] = __int_as_float(z);
}

// The butterfly up to its row-of-16 stage, folded into a RUNNING maximum: zrun's lane l holds, for accumulator register
// (l & 15), the maximum over the row of 16 lanes and over every 32-position slice passed so far.  A wave that walks all K
// neighbours of a centre itself (K / 32 slices) needs no LDS and no barrier for the max: sw_bfly_finish joins the two rows
// of a half-wave once per centre.  zrun starts at 0, which is the ReLU.
__device__ __forceinline__ void sw_bfly_accumulate(const f32x16 &acc, int &zrun, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    int w[8], v[4], u[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int x0 = __float_as_int(acc[2 * i]), x1 = __float_as_int(acc[2 * i + 1]);
        const int own = b0 ? x1 : x0, oth = b0 ? x0 : x1;
        w[i] = sw_imax_dpp<0xB1>(own, oth);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int own = b1 ? w[2 * j + 1] : w[2 * j], oth = b1 ? w[2 * j] : w[2 * j + 1];
        v[j] = sw_imax_dpp<0x4E>(own, oth);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int own = b2 ? v[2 * m + 1] : v[2 * m], oth = b2 ? v[2 * m] : v[2 * m + 1];
        const int p = sw_dpp_sel<0x104, 0x5>(own, oth);
        const int q = sw_dpp_sel<0x114, 0xA>(p, oth);
        u[m] = sw_imax(own, q);
    }
    const int own = b3 ? u[1] : u[0], oth = b3 ? u[0] : u[1];
    zrun = sw_imax(zrun, sw_imax_dpp<0x128>(own, oth));
}
// the finished maximum of output tile t's rows: valid in the lanes with (lane & 16) == 0, lane l = row
// 32 t + 8 ((l & 15) >> 2) + (l & 3) + 4 (l >> 5)
__device__ __forceinline__ float sw_bfly_finish(int zrun) {
    return __int_as_float(sw_imax(zrun, __builtin_amdgcn_ds_swizzle(zrun, 0x401F)));     // lane ^ 16
}


// One layer whose input activations are B-operand registers hin[].  START = parity of the register set
// that holds this layer's first weight set (loaded by the previous phase); `next` loads the following
// layer's first set into the set after this layer's last one.
enum { SW_EPI_MID = 0, SW_EPI_MAX = 1, SW_EPI_STORE = 2 };

// destination of a SW_EPI_STORE layer: y points at this wave's first column of output row 0
struct SwStore {
    float *y = nullptr;
    long long ld = 0;   // elements between output rows
    bool col_ok = true; // this lane's column exists
    int act = 1;        // ACT_* of common.h
};

// act + store of output tile t: rows 32t.. of y[row*ld + lane&31]
template <int COUT>
__device__ __forceinline__ void sw_store_epilogue(const f32x16 &acc, int t, const SwStore &st, int lane) {
    if (!st.col_ok) return;
    float *yp = st.y + (size_t)(32 * t + 4 * (lane >> 5)) * st.ld + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        if (32 * t + ro + 4 < COUT || 32 * t + ro + 4 * (lane >> 5) < COUT) yp[(size_t)ro * st.ld] = apply_act(acc[r], st.act);
    }
}

template <int CIN, int COUT, int EPI, int START, int NIN, int NOUT, typename Next>
__device__ __forceinline__ void sw_layer_reg(const float *wt, const float *bias_lds, const float (&hin)[NIN], float (&hout)[NOUT],
                                             float (&s)[2][16], float *red, int wave, int lane, Next next, const SwStore &st = SwStore()) {
    using S = SwShape<CIN, COUT>;
    static_assert(NIN >= S::KST, "input operand array too small");
    const __amdgpu_buffer_rsrc_t rsrc = sw_frag_rsrc<CIN, COUT>(wt);
    const int voff = lane * 16;
    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < S::STEPS; ++g) {
        const int ps = g / S::NSETS, c = g % S::NSETS;
        if (c == 0) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                if (2 * ps + tm < S::NT) sw_bias_init(acc[tm], bias_lds, 2 * ps + tm, lane);
        }
        if (g + 1 < S::STEPS) {
            sw_load_set<CIN, COUT>(s[(START + g + 1) & 1], rsrc, voff, (g + 1) / S::NSETS, (g + 1) % S::NSETS);
        } else {
            next(s[(START + g + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < SW_KS; ++j)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int kk = c * SW_KS + j;
                if (kk < S::KST && 2 * ps + tm < S::NT)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[(START + g) & 1][tm * SW_KS + j], hin[kk], acc[tm], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (c == S::NSETS - 1) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                if (2 * ps + tm < S::NT) {
                    if (EPI == SW_EPI_MAX) sw_last_epilogue_bfly<COUT>(acc[tm], 2 * ps + tm, red, wave, lane);
                    else if (EPI == SW_EPI_STORE) sw_store_epilogue<COUT>(acc[tm], 2 * ps + tm, st, lane);
                    else sw_mid_epilogue<NOUT>(acc[tm], 2 * ps + tm, hout);
                }
        }
    }
}

}  // namespace
