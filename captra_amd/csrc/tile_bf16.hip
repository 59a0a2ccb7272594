// bf16-native dense layers, second generation (round 4): the operand TILE in LDS, every weight fragment feeding FOUR position
// tiles.  Replaces, for the wide layers of the rotation heads (reference network/models/blocks.py:147-193: Conv1d 1x1 ->
// GroupNorm -> ReLU chains on 4096 points per cloud) and every other point-major layer with >= 64 output channels, the
// streaming kernels of csrc/dense_bf16.hip, whose wave tile (128 rows x 64 positions) re-read every 1 KiB weight fragment
// for two MFMAs only -- 64 B/clk per CU of L1 traffic at the matrix pipe's rate, exactly what the L1 delivers -- and fetched
// the activations as fragment-shaped 16-byte-per-row loads (32 cache lines per instruction).
//
//   tb_layer_kernel<MT, NW>: a workgroup of NW waves owns 128 positions x (NW x MT x 32) output channels.  The activations
//     (B, L, ceil32(C)) bf16 point-major, slot order (csrc/dense_bf16.hip header) are staged through LDS in K-chunks of 128
//     channels (32 KiB, double-buffered, one barrier per chunk) with fully coalesced 16-byte loads; the GroupNorm of the
//     producing layer, x -> bf16(relu(a x + b)), is applied ONCE per element while it is staged.  The LDS image is
//     [position][16 slots of 16 bytes] with slot ^= position & 15, so the B fragments (one ds_read_b128 per lane: the 16-byte
//     slot 2 kk + h of position col) and the staging stores are bank-conflict free.  A wave carries MT x 4 accumulator tiles:
//     per k-step MT weight-fragment loads (fragment image, ring of four k-steps), four B reads, 4 MT MFMAs.
//   tb_head12_kernel: the first two layers of a rotation head in one launch -- y1 = W1 x + b1 (128 -> 512) is recomputed from
//     the 32 KiB x tile instead of being written to HBM and read back (134 + 134 MB per step at 32 x 4096 points): MODE 0 is
//     the statistics pass (y1's group sums, nothing stored), MODE 1 computes y1 again, applies its GroupNorm + ReLU to the
//     ACCUMULATORS, parks bf16 y1 as the [128 positions][512 channels] LDS image (128 KiB) and runs the 512 -> 512 layer from
//     there; its epilogue stores raw y2 and y2's partial statistics.
//
// Contract per layer (unchanged): y = act(b + sum_k bf16(w[k]) * bf16(x[k])), fp32 accumulation, k ascending.  Statistics are
// partial (sum, sum of squares) per channel and chunk of 128 positions, of the fp32 accumulator values, tile-major
// (B, T, C, 2) for captra_gn_finalize_tm; fixed summation order, no atomics.
#include "common.h"
#include "bf16_dense.h"
#include <atomic>

namespace {

constexpr int TB_P = 128;          // positions per workgroup
constexpr int TB_TN = 4;           // 32-position tiles per wave
constexpr int TB_CHUNK = 32768;    // one staged K-chunk: 128 positions x 128 channels bf16

struct TbParams {
    int cin, cout, kst, nt, cp_in, cp_out;
    long long L;
    const unsigned char *x;        // IN 0: (B,L,cp_in) bf16 slot order; IN 1: (B,csplit,L) fp32 channel-major (channels < csplit)
    const float *x2;               // IN 1: channels >= csplit, (B,cin - csplit,L) fp32 (the concat [x, x2] is never built)
    int csplit;
    const unsigned char *wimg;     // captra_pack_dense_bf16(perm = 1)
    const float *bias;
    long long bias_bs;             // per-cloud bias stride (0: one bias)
    const float *ab;               // AFF: (B,cin,2)
    unsigned char *y;              // OUT 0: (B,L,cp_out) bf16 slot order; OUT 1: (B,cout,L) fp32; OUT 2: (B,cout) fp32 = max over the positions
    int act;
    float *stats;                  // ST: (B,st_t,cout,2)
    int st_t;
};

// x -> bf16(relu(a x + b)) on the 8 channels of one 16-byte slot
__device__ __forceinline__ u32x4 tb_affine(u32x4 v, const float (&a)[8], const float (&b)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(v[i] << 16), hi = __uint_as_float(v[i] & 0xffff0000u);
        r[i] = db_relu2(db_pack(__builtin_fmaf(lo, a[2 * i], b[2 * i]), __builtin_fmaf(hi, a[2 * i + 1], b[2 * i + 1])));
    }
    return r;
}

// store one accumulator tile as two 16-byte slots of a point-major row (+ ReLU), global memory
__device__ __forceinline__ void tb_store_tile(const f32x16 &acc, unsigned char *yrow, bool relu) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = db_pack(acc[8 * jj + 2 * i], acc[8 * jj + 2 * i + 1]);
            if (relu) v[i] = db_relu2(v[i]);
        }
        *reinterpret_cast<u32x4 *>(yrow + 32 * jj) = v;
    }
}
__device__ __forceinline__ void tb_stats_acc(float (&sv)[32], const f32x16 &acc, bool valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = valid ? acc[r] : 0.f;
        sv[r] += v;
        sv[16 + r] = __builtin_fmaf(v, v, sv[16 + r]);
    }
}

template <int MT, int NW, int IN, bool AFF, int OUT, bool ST, bool KF>
__global__ __launch_bounds__(NW * 64, 2) void tb_layer_kernel(TbParams p) {
    static_assert(!(AFF && IN == 1) && !(ST && OUT != 0), "GroupNorm on load / statistics are point-major features");
    constexpr int NT = NW * 64, UN = 2048 / NT, PSTEP = NT / 16;
    constexpr int ITEMS = 512 / NT;                                   // IN 1: (position quad, slot) items per thread and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *aff_tab = reinterpret_cast<float *>(lds + 2 * TB_CHUNK);   // AFF: [2 kst slots][a0..a7, b0..b7]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int b = blockIdx.z;
    const int t0 = (blockIdx.y * NW + wave) * MT;
    const long long pos0 = (long long)blockIdx.x * TB_P;
    const int kst = p.kst, nch = (kst + 7) >> 3;
    // ---- staging geometry: this thread moves the 16-byte slot `sslot` of positions spos, spos + PSTEP, ... ----------
    const int sslot = tid & 15, spos = tid >> 4;
    const unsigned char *xb = p.x + (IN == 0 ? (size_t)b * p.L * p.cp_in * 2 : 0);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)(IN == 0 ? p.L * p.cp_in * 2 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.wimg, 0, p.nt * kst * 1024, 0x00020000);
    int xoff[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        long long c = pos0 + spos + i * PSTEP;
        if (c >= p.L) c = p.L - 1;                                   // clamped position: computed, never stored
        xoff[i] = (int)(c * p.cp_in * 2) + sslot * 16;
    }
    const int lw0 = spos * 256 + ((sslot ^ (spos & 15)) << 4);       // (spos + i PSTEP) & 15 == spos & 15
    if constexpr (AFF) {
        const float *abp = p.ab + (size_t)b * p.cin * 2;
        for (int e = tid; e < kst * 32; e += NT) {
            const int kk = e >> 5, hh = (e >> 4) & 1, i = e & 15;
            const int c = 16 * kk + db_perm(8 * hh + (i & 7));
            aff_tab[e] = c < p.cin ? abp[2 * c + (i >> 3)] : 0.f;
        }
    }
    int woff[MT];
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) woff[tm] = (t0 + tm < p.nt ? t0 + tm : p.nt - 1) * kst * 1024;   // clamped row tile: computed, never stored
    f32x16 acc[MT][TB_TN];
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        const float *bp = p.bias + (size_t)b * p.bias_bs + (t0 + tm < p.nt ? t0 + tm : p.nt - 1) * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int tn = 0; tn < TB_TN; ++tn) acc[tm][tn][r] = bv;
        }
    }
    u32x4 A[4][MT];                                                   // k-step kk lives in set kk & 3, loaded three k-steps ahead
    auto loadA = [&](int s, int kk) {
        kk = kk < kst ? kk : kst - 1;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, woff[tm] + kk * 1024, 0);
    };
    u32x4 st[IN == 0 ? UN : 1];                                       // IN 0: the NEXT chunk's units, in flight / waiting to be parked
    float4 sf[IN == 1 ? ITEMS : 1][8];                                // IN 1: 8 channels x 4 positions per item
    // IN 1 geometry: item u = tid + i NT -> position quad u & 31, slot u >> 5 of the chunk; the slot's 8 channels are
    // base + {0,1,2,3,8,9,10,11}, base = 16 kk + 4 hh (slot order); a channel-major fp32 source, float4 = 4 positions
    const float *x0b = reinterpret_cast<const float *>(p.x) + (IN == 1 ? (size_t)b * p.csplit * p.L : 0);
    const float *x1b = p.x2 + (IN == 1 ? (size_t)b * (p.cin - p.csplit) * p.L : 0);
    auto gload = [&](int c) {
        c = c < nch ? c : nch - 1;                                    // (always issued: the load counts stay static)
        if constexpr (IN == 0) {
#pragma unroll
            for (int i = 0; i < UN; ++i) st[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[i], c * 256, 0);
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int u = tid + i * NT, pq = u & 31, sl = u >> 5;
                long long pos = pos0 + 4 * pq;
                if (pos > p.L - 4) pos = p.L - 4;                     // clamped quad: computed, never stored
                const int base = 16 * (8 * c + (sl >> 1)) + 4 * (sl & 1);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int ch = base + (e & 3) + 8 * (e >> 2);
                    ch = ch < p.cin ? ch : p.cin - 1;                 // clamped channel: loaded, zeroed when parked
                    const float *src = ch < p.csplit ? x0b + (size_t)ch * p.L : x1b + (size_t)(ch - p.csplit) * p.L;
                    sf[i][e] = *reinterpret_cast<const float4 *>(src + pos);
                }
            }
        }
    };
    auto park = [&](int c) {
        if constexpr (IN == 1) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int u = tid + i * NT, pq = u & 31, sl = u >> 5;
                const int base = 16 * (8 * c + (sl >> 1)) + 4 * (sl & 1);
                float m[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = base + (e & 3) + 8 * (e >> 2) < p.cin ? 1.f : 0.f;
                unsigned char *dst = lds + (c & 1) * TB_CHUNK;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pos = 4 * pq + r;
                    u32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = r == 0 ? sf[i][2 * q].x : r == 1 ? sf[i][2 * q].y : r == 2 ? sf[i][2 * q].z : sf[i][2 * q].w;
                        const float hi = r == 0 ? sf[i][2 * q + 1].x : r == 1 ? sf[i][2 * q + 1].y : r == 2 ? sf[i][2 * q + 1].z : sf[i][2 * q + 1].w;
                        v[q] = db_pack(m[2 * q] != 0.f ? lo : 0.f, m[2 * q + 1] != 0.f ? hi : 0.f);
                    }
                    *reinterpret_cast<u32x4 *>(dst + pos * 256 + ((sl ^ (pos & 15)) << 4)) = v;
                }
            }
            return;
        }
        unsigned char *dst = lds + (c & 1) * TB_CHUNK + lw0;
        if constexpr (AFF) {
            const int s = 16 * c + sslot;
            const float4 *t = reinterpret_cast<const float4 *>(aff_tab + (s < 2 * kst ? s : 2 * kst - 1) * 16);
            const float4 a0 = t[0], a1 = t[1], b0 = t[2], b1 = t[3];
            const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < UN; ++i) *reinterpret_cast<u32x4 *>(dst + i * PSTEP * 256) = tb_affine(st[i], a, bb);
        } else if constexpr (IN == 0) {
#pragma unroll
            for (int i = 0; i < UN; ++i) *reinterpret_cast<u32x4 *>(dst + i * PSTEP * 256) = st[i];
        }
    };
    gload(0);
    loadA(0, 0);
    loadA(1, 1);
    loadA(2, 2);
    if constexpr (AFF) __syncthreads();                               // the coefficient table
    park(0);
    gload(1);
    __syncthreads();
    const int e0 = (h ^ (col & 15)) << 4;
    const int brd = col * 256;
    // Per k-step: the weight fragments of k-step kk + 3 are requested, the B fragments of k-step kk + 1 read (two register
    // sets), then the 4 MT MFMAs of k-step kk issue.  sched_barrier pins that order: left alone, the scheduler sinks the
    // fragment loads to one k-step before their use and every k-step waits for an L2 round trip.
    u32x4 Bf[2][TB_TN];
    for (int c = 0; c < nch; ++c) {
        const unsigned char *src = lds + (c & 1) * TB_CHUNK + brd;
#pragma unroll
        for (int tn = 0; tn < TB_TN; ++tn) Bf[0][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * c + j;
            loadA((j + 3) & 3, kk + 3);
            if (j < 7 && (KF || kk + 1 < kst)) {
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn) Bf[(j + 1) & 1][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + ((32 * (j + 1)) ^ e0));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (KF || kk < kst) {
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = db_mfma(A[j & 3][tm], Bf[j & 1][tn], acc[tm][tn]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j == 1 && c + 1 < nch) {                              // the next chunk: transform + park under this chunk's MFMAs
                park(c + 1);
                gload(c + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue ------------------------------------------------------------------------------------------------
    const bool relu = p.act == ACT_RELU;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        const int t = t0 + tm;
        if (t >= p.nt) continue;
        float sv[ST ? 32 : 1];
        if constexpr (ST) {
#pragma unroll
            for (int i = 0; i < 32; ++i) sv[i] = 0.f;
        }
        if constexpr (OUT == 2) {
            // max over the workgroup's positions (the launcher guarantees one position tile per cloud): lane-local over the
            // four tiles, then a 32-column butterfly; act(max) == max(act) for the monotone activations
            float mx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mx[r] = -INFINITY;
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn) mx[r] = pos0 + tn * 32 + col < p.L ? fmaxf(mx[r], acc[tm][tn][r]) : mx[r];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], off, 64));
            }
            if (col == 0) {
                float *yp = reinterpret_cast<float *>(p.y) + (size_t)b * p.cout + 32 * t + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (32 * t + 4 * h + ro < p.cout) yp[ro] = apply_act(mx[r], p.act);
                }
            }
            continue;
        }
#pragma unroll
        for (int tn = 0; tn < TB_TN; ++tn) {
            const long long c = pos0 + tn * 32 + col;
            const bool valid = c < p.L;
            if constexpr (OUT == 0) {
                if (valid) tb_store_tile(acc[tm][tn], p.y + (((size_t)b * p.L + c) * p.cp_out + 32 * t + 8 * h) * 2, relu);
            } else if constexpr (OUT == 1) {
                const int row0 = 32 * t + 4 * h;
                float *yp = reinterpret_cast<float *>(p.y) + ((size_t)b * p.cout + row0) * p.L + c;
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        if (row0 + ro < p.cout) yp[(size_t)ro * p.L] = apply_act(acc[tm][tn][r], p.act);
                    }
                }
            }
            if constexpr (ST) tb_stats_acc(sv, acc[tm][tn], valid);
        }
        if constexpr (ST) db_stats_tile(sv, col, h, t, p.cout, p.stats + (size_t)b * p.cout * p.st_t * 2, p.st_t, (int)blockIdx.x);
    }
}

// ---- rotation head, layers 1 + 2 in one launch ----------------------------------------------------------------------------
struct HbParams {
    int cin, kst1, cp_x;           // layer 1: cin <= 128 input channels (kst1 <= 8), x rows of cp_x elements
    long long L;
    const unsigned char *x;        // (B,L,cp_x) bf16 slot order
    const unsigned char *w1;       // 128 -> 512 fragment image (nt = 16)
    const float *bias1;
    const float *ab1;              // MODE 1: (B,512,2) GroupNorm coefficients of y1
    const unsigned char *w2;       // 512 -> 512 fragment image (kst = 32, nt = 16)
    const float *bias2;
    unsigned char *y2;             // MODE 1: (B,L,512) bf16 slot order, raw
    float *stats;                  // MODE 0: y1's partial statistics; MODE 1: y2's; (B,st_t,512,2)
    int st_t;
    int dbg;
};

template <int MODE>
__global__ __launch_bounds__(512, 2) void tb_head12_kernel(HbParams p) {
    constexpr int MT = 2, C1 = 512;
    constexpr int XOFF = MODE == 1 ? 98304 : 0;                       // the x tile lives in the tail of the y1 image until y1 is written
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int b = blockIdx.y;
    const int t0 = wave * MT;
    const long long pos0 = (long long)blockIdx.x * TB_P;
    const int kst1 = p.kst1;
    const __amdgpu_buffer_rsrc_t w1src = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, 16 * kst1 * 1024, 0x00020000);
    // ---- x tile -> LDS (row pitch 256 bytes, slot ^= position & 15) ---------------------------------------------------
    {
        const int sslot = tid & 15, spos = tid >> 4;
        const unsigned char *xb = p.x + (size_t)b * p.L * p.cp_x * 2;
        const int nsl = p.cp_x >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long c = pos0 + spos + i * 32;
            if (c >= p.L) c = p.L - 1;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (sslot < nsl) v = *reinterpret_cast<const u32x4 *>(xb + (size_t)c * p.cp_x * 2 + sslot * 16);
            *reinterpret_cast<u32x4 *>(lds + XOFF + (spos + i * 32) * 256 + ((sslot ^ (spos & 15)) << 4)) = v;
        }
    }
    f32x16 acc[MT][TB_TN];
    auto init_acc = [&](const float *bias) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const float *bp = bias + (t0 + tm) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn) acc[tm][tn][r] = bv;
            }
        }
    };
    u32x4 A[4][MT];
    auto loadA1 = [&](int s, int kk) {
        kk = kk < kst1 ? kk : kst1 - 1;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(w1src, lane * 16, ((t0 + tm) * kst1 + kk) * 1024, 0);
    };
    init_acc(p.bias1);
    loadA1(0, 0);
    loadA1(1, 1);
    loadA1(2, 2);
    __syncthreads();
    const int e0 = (h ^ (col & 15)) << 4;
    u32x4 Bf[2][TB_TN];
    {
        const unsigned char *src = lds + XOFF + col * 256;
#pragma unroll
        for (int tn = 0; tn < TB_TN; ++tn) Bf[0][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            loadA1((j + 3) & 3, j + 3);
            if (j < 7 && j + 1 < kst1) {
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn) Bf[(j + 1) & 1][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + ((32 * (j + 1)) ^ e0));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j < kst1) {
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = db_mfma(A[j & 3][tm], Bf[j & 1][tn], acc[tm][tn]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (MODE == 0) {
        // ---- statistics pass: y1's partial group sums, nothing stored ---------------------------------------------------
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            float sv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) sv[i] = 0.f;
#pragma unroll
            for (int tn = 0; tn < TB_TN; ++tn) tb_stats_acc(sv, acc[tm][tn], pos0 + tn * 32 + col < p.L);
            db_stats_tile(sv, col, h, t0 + tm, C1, p.stats + (size_t)b * C1 * p.st_t * 2, p.st_t, (int)blockIdx.x);
        }
        return;
    } else {
        const __amdgpu_buffer_rsrc_t w2src = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2, 0, 16 * 32 * 1024, 0x00020000);
        auto loadA2 = [&](int s, int kk) {
            kk = kk < 32 ? kk : 31;
            if (p.dbg & 1) kk = 0;
            if (p.dbg & 16) return;
#pragma unroll
            for (int tm = 0; tm < MT; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(w2src, lane * 16, ((t0 + tm) * 32 + kk) * 1024, 0);
        };
        loadA2(0, 0);                                                 // (layer 2's first fragments under the hand-over)
        loadA2(1, 1);
        loadA2(2, 2);
        __syncthreads();                                              // every wave is done with the x tile
        // ---- y1 -> bf16(relu(a y1 + b)) -> the [128][512] LDS image (slot 4 t + 2 jj + h of position col) -------------
        const float *abp = p.ab1 + (size_t)b * C1 * 2;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const int t = t0 + tm;
            float ca[16], cb[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                             // rows 8 q + 4 h + {0..3} of the tile: 8 floats (a, b interleaved) each
                const float4 *s4 = reinterpret_cast<const float4 *>(abp + (32 * t + 8 * q + 4 * h) * 2);
                const float4 u0 = s4[0], u1 = s4[1];
                ca[4 * q + 0] = u0.x; cb[4 * q + 0] = u0.y; ca[4 * q + 1] = u0.z; cb[4 * q + 1] = u0.w;
                ca[4 * q + 2] = u1.x; cb[4 * q + 2] = u1.y; ca[4 * q + 3] = u1.z; cb[4 * q + 3] = u1.w;
            }
#pragma unroll
            for (int tn = 0; tn < TB_TN; ++tn) {
                unsigned char *row = lds + (tn * 32 + col) * 1024;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    u32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 8 * jj + 2 * i;
                        v[i] = db_relu2(db_pack(__builtin_fmaf(acc[tm][tn][r], ca[r], cb[r]), __builtin_fmaf(acc[tm][tn][r + 1], ca[r + 1], cb[r + 1])));
                    }
                    *reinterpret_cast<u32x4 *>(row + (((4 * t + 2 * jj + h) ^ (col & 15)) << 4)) = v;
                }
            }
        }
        init_acc(p.bias2);
        __syncthreads();
        // ---- layer 2 from the LDS image ----------------------------------------------------------------------------------
        const unsigned char *src = lds + col * 1024;
#pragma unroll
        for (int tn = 0; tn < TB_TN; ++tn) Bf[0][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 32768 + e0);
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 8 * c + j;
                loadA2((j + 3) & 3, kk + 3);
                {
                    const int kn = kk + 1 < 32 ? kk + 1 : 31;          // (the last k-step re-reads itself: never multiplied)
                    const int cn = kn >> 3, jn = (j + 1) & 7;
                    if (!(p.dbg & 8))
#pragma unroll
                    for (int tn = 0; tn < TB_TN; ++tn)
                        Bf[(j + 1) & 1][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 32768 + cn * 256 + ((32 * jn) ^ e0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = db_mfma(A[j & 3][tm], Bf[j & 1][tn], acc[tm][tn]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const int t = t0 + tm;
            float sv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) sv[i] = 0.f;
#pragma unroll
            for (int tn = 0; tn < TB_TN; ++tn) {
                const long long c = pos0 + tn * 32 + col;
                const bool valid = c < p.L;
                if (valid && !(p.dbg & 2)) tb_store_tile(acc[tm][tn], p.y2 + (((size_t)b * p.L + c) * C1 + 32 * t + 8 * h) * 2, false);
                if (!(p.dbg & 4)) tb_stats_acc(sv, acc[tm][tn], valid);
            }
            if (!(p.dbg & 4)) db_stats_tile(sv, col, h, t, C1, p.stats + (size_t)b * C1 * p.st_t * 2, p.st_t, (int)blockIdx.x);
        }
    }
}

// ---- the same pair, PERSISTENT: a workgroup per CU walks a contiguous run of 128-position tiles ----------------------------
// tb_head12_kernel<1> pays, per tile: the x tile's HBM latency at workgroup start, the bias loads of both layers, the
// GroupNorm coefficients' loads in the hand-over (all on the critical path of eight waves that meet at the same barriers), and
// a launch ramp per round of workgroups (four rounds at 32 x 4096 points) -- 30 of its 99 us (ablation).  Here the
// biases and the cloud's coefficients live in LDS (reloaded when the run crosses into another cloud), the NEXT tile's x rows are
// requested right after the hand-over and travel under layer 2's 256 MFMAs per wave, and the kernel has one ramp.
struct HpParams {
    HbParams q;
    int ntiles, tpc, tpw;          // tiles in the launch, tiles per cloud, tiles per workgroup (contiguous runs)
};

__global__ __launch_bounds__(512, 2) void tb_head12p_kernel(HpParams pp) {
    const HbParams &p = pp.q;
    constexpr int MT = 2, C1 = 512;
    constexpr int XOFF = 98304, TAB = 131072;                          // x tile in the tail of the y1 image; tables behind it
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *ab_l = reinterpret_cast<float *>(lds + TAB);                // [512][2] coefficients of the current cloud
    float *b1_l = ab_l + 1024, *b2_l = b1_l + 512;                     // [512] biases of layers 1 and 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int t0 = wave * MT;
    const int kst1 = p.kst1;
    const __amdgpu_buffer_rsrc_t w1src = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, 16 * kst1 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2src = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2, 0, 16 * 32 * 1024, 0x00020000);
    const int tile0 = blockIdx.x * pp.tpw;
    const int tile1 = tile0 + pp.tpw < pp.ntiles ? tile0 + pp.tpw : pp.ntiles;
    if (tile0 >= tile1) return;
    for (int e = tid; e < 512; e += 512) { b1_l[e] = p.bias1[e]; b2_l[e] = p.bias2[e]; }
    __syncthreads();
    // x tile staging: thread -> 16-byte slot sslot of positions spos + 32 i
    const int sslot = tid & 15, spos = tid >> 4, nsl = p.cp_x >> 3;
    u32x4 xr[4];
    auto xload = [&](int tile) {
        const int b = tile / pp.tpc;
        const long long pos0 = (long long)(tile - b * pp.tpc) * TB_P;
        const unsigned char *xb = p.x + (size_t)b * p.L * p.cp_x * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long c = pos0 + spos + i * 32;
            if (c >= p.L) c = p.L - 1;
            xr[i] = u32x4{0u, 0u, 0u, 0u};
            if (sslot < nsl) xr[i] = *reinterpret_cast<const u32x4 *>(xb + (size_t)c * p.cp_x * 2 + sslot * 16);
        }
    };
    auto xpark = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4 *>(lds + XOFF + (spos + i * 32) * 256 + ((sslot ^ (spos & 15)) << 4)) = xr[i];
    };
    f32x16 acc[MT][TB_TN];
    auto init_acc = [&](const float *bias_l) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const float *bp = bias_l + (t0 + tm) * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4 *>(bp + 8 * q);
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn) { acc[tm][tn][4 * q] = bv.x; acc[tm][tn][4 * q + 1] = bv.y; acc[tm][tn][4 * q + 2] = bv.z; acc[tm][tn][4 * q + 3] = bv.w; }
            }
        }
    };
    u32x4 A[4][MT];
    auto loadA1 = [&](int s, int kk) {
        kk = kk < kst1 ? kk : kst1 - 1;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(w1src, lane * 16, ((t0 + tm) * kst1 + kk) * 1024, 0);
    };
    auto loadA2 = [&](int s, int kk) {
        kk = kk < 32 ? kk : 31;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(w2src, lane * 16, ((t0 + tm) * 32 + kk) * 1024, 0);
    };
    const int e0 = (h ^ (col & 15)) << 4;
    xload(tile0);
    xpark();
    int cur_b = -1;
    for (int tile = tile0; tile < tile1; ++tile) {
        const int b = tile / pp.tpc;
        const long long pos0 = (long long)(tile - b * pp.tpc) * TB_P;
        if (b != cur_b) {                                              // (workgroup-uniform) the cloud's GroupNorm coefficients
            const float *abp = p.ab1 + (size_t)b * C1 * 2;
            for (int e = tid; e < 1024; e += 512) ab_l[e] = abp[e];
            cur_b = b;
        }
        init_acc(b1_l);
        loadA1(0, 0);
        loadA1(1, 1);
        loadA1(2, 2);
        __syncthreads();                                               // A: the x tile (and the tables) are in LDS
        {
            const unsigned char *src = lds + XOFF + col * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                loadA1((j + 3) & 3, j + 3);
                if (j < kst1) {
                    u32x4 Bf[TB_TN];
#pragma unroll
                    for (int tn = 0; tn < TB_TN; ++tn) Bf[tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + ((32 * j) ^ e0));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tn = 0; tn < TB_TN; ++tn)
#pragma unroll
                        for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = db_mfma(A[j & 3][tm], Bf[tn], acc[tm][tn]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        loadA2(0, 0);
        loadA2(1, 1);
        loadA2(2, 2);
        __syncthreads();                                               // B: every wave is done with the x tile
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const int t = t0 + tm;
            float ca[16], cb[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 *s4 = reinterpret_cast<const float4 *>(ab_l + (32 * t + 8 * q + 4 * h) * 2);
                const float4 u0 = s4[0], u1 = s4[1];
                ca[4 * q + 0] = u0.x; cb[4 * q + 0] = u0.y; ca[4 * q + 1] = u0.z; cb[4 * q + 1] = u0.w;
                ca[4 * q + 2] = u1.x; cb[4 * q + 2] = u1.y; ca[4 * q + 3] = u1.z; cb[4 * q + 3] = u1.w;
            }
#pragma unroll
            for (int tn = 0; tn < TB_TN; ++tn) {
                unsigned char *row = lds + (tn * 32 + col) * 1024;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    u32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 8 * jj + 2 * i;
                        v[i] = db_relu2(db_pack(__builtin_fmaf(acc[tm][tn][r], ca[r], cb[r]), __builtin_fmaf(acc[tm][tn][r + 1], ca[r + 1], cb[r + 1])));
                    }
                    *reinterpret_cast<u32x4 *>(row + (((4 * t + 2 * jj + h) ^ (col & 15)) << 4)) = v;
                }
            }
        }
        init_acc(b2_l);
        __syncthreads();                                               // C: y1 is in LDS
        const unsigned char *src = lds + col * 1024;
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 8 * c + j;
                loadA2((j + 3) & 3, kk + 3);
                u32x4 Bf[TB_TN];
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn) Bf[tn] = *reinterpret_cast<const u32x4 *>(src + tn * 32768 + c * 256 + ((32 * j) ^ e0));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tn = 0; tn < TB_TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = db_mfma(A[j & 3][tm], Bf[tn], acc[tm][tn]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next tile's x rows, requested BEHIND the last weight fragment this tile needs (the load counter retires in order: issued
        // earlier, every later fragment wait would also wait for this HBM round trip) and once the operand registers are free (with
        // them live the kernel spills); they travel under the epilogue's stores and statistics
        if (tile + 1 < tile1) xload(tile + 1);
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const int t = t0 + tm;
            float sv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) sv[i] = 0.f;
#pragma unroll
            for (int tn = 0; tn < TB_TN; ++tn) {
                const long long c = pos0 + tn * 32 + col;
                const bool valid = c < p.L;
                if (valid) tb_store_tile(acc[tm][tn], p.y2 + (((size_t)b * p.L + c) * C1 + 32 * t + 8 * h) * 2, false);
                tb_stats_acc(sv, acc[tm][tn], valid);
            }
            db_stats_tile(sv, col, h, t, C1, p.stats + (size_t)b * C1 * p.st_t * 2, p.st_t, (int)(tile - b * pp.tpc));
        }
        __syncthreads();                                               // D: every wave is done with y1
        if (tile + 1 < tile1) xpark();
    }
}

template <int MT, int NW, int IN, bool AFF, int OUT, bool ST>
int tb_launch(int b, const TbParams &p, hipStream_t s) {
    dim3 grid((unsigned)((p.L + TB_P - 1) / TB_P), (p.nt + NW * MT - 1) / (NW * MT), b);
    const int lds = 2 * TB_CHUNK + (AFF ? p.kst * 128 : 0);
    static CaptraDeviceOnce once_f, once_p;
    constexpr bool CAN_KF = IN == 0 && OUT == 0;
    if (CAN_KF && p.kst % 8 == 0) {
        if (once_f.first_use()) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(&tb_layer_kernel<MT, NW, IN, AFF, OUT, ST, CAN_KF>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_CHUNK + 65536);
            once_f.done();
        }
        CAPTRA_LAUNCH("pointwise_mlp", (tb_layer_kernel<MT, NW, IN, AFF, OUT, ST, CAN_KF>), grid, dim3(NW * 64), lds, s, p);
    } else {
        if (once_p.first_use()) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(&tb_layer_kernel<MT, NW, IN, AFF, OUT, ST, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_CHUNK + 65536);
            once_p.done();
        }
        CAPTRA_LAUNCH("pointwise_mlp", (tb_layer_kernel<MT, NW, IN, AFF, OUT, ST, false>), grid, dim3(NW * 64), lds, s, p);
    }
    return captra_last_error();
}

// ---- one vector per cloud through a layer: y[b] = bias + W bf16(x[b]) ---------------------------------------------------
// (FP3's per-cloud bias W2 v + b, pointnet_utils.py:265-268 with S == 1.)  A lane owns FOUR consecutive output channels (16-byte
// loads of the row-major W'^T rows), the sixteen waves split k as k = 16 i + wave -- 64 dependent steps for cin = 1024, sixteen
// loads in flight per wave -- and their partial sums are added in wave order.
__global__ __launch_bounds__(1024) void tb_gemv_kernel(int cin, int cout, int ldw, const float *__restrict__ x, const float *__restrict__ wt,
                                                        const float *__restrict__ bias, float *__restrict__ y) {
    __shared__ float4 red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = (blockIdx.x * 64 + lane) * 4, b = blockIdx.y;
    const int cc = co < ldw ? co : ldw - 4;                            // (ldw = ceil128(cout): whole float4s, zero padded)
    const float *xb = x + (size_t)b * cin;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (int k = wave; k < cin; k += 16) {
        const float xv = (float)(__bf16)xb[k];
        const float4 w = *reinterpret_cast<const float4 *>(wt + (size_t)k * ldw + cc);
        acc.x = __builtin_fmaf((float)(__bf16)w.x, xv, acc.x);
        acc.y = __builtin_fmaf((float)(__bf16)w.y, xv, acc.y);
        acc.z = __builtin_fmaf((float)(__bf16)w.z, xv, acc.z);
        acc.w = __builtin_fmaf((float)(__bf16)w.w, xv, acc.w);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float4 s = red[0][lane];
        for (int w = 1; w < 16; ++w) {
            const float4 t = red[w][lane];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float v[4] = {s.x, s.y, s.z, s.w};
        for (int i = 0; i < 4; ++i)
            if (co + i < cout) y[(size_t)b * cout + co + i] = bias[co + i] + v[i];
    }
}

}  // namespace

// x (B,cin) fp32, wt_packed / bias_packed: the layer's packed fp32 buffers -> y (B,cout) fp32 = bias + sum_k bf16(w[k]) bf16(x[k]).
extern "C" int captra_gemv_bf16(int b, int cin, int cout, const float *x, const float *wt_packed, const float *bias_packed, float *y,
                                captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("pointwise_mlp", tb_gemv_kernel, dim3((cout + 255) / 256, b), dim3(1024), 0, (hipStream_t)stream, cin, cout,
                  (cout + 127) / 128 * 128, x, wt_packed, bias_packed, y);
    return captra_last_error();
}

static CAPTRA_KNOB int g_tb_dbg = 0;
extern "C" void captra_tile_bf16_set_debug(int v) { g_tb_dbg = v; }
static CAPTRA_KNOB int g_tb_persist = 1;       // fused head pair: 1 = the persistent form when the launch has more tiles than CUs
extern "C" void captra_tile_bf16_set_persistent(int v) { g_tb_persist = v; }

extern "C" int captra_dense_bf16_tile_stats_tiles(long long l) { return (int)((l + TB_P - 1) / TB_P); }

// One dense layer through the LDS-tiled kernel.
//   in_cm = 0: x (B,L,ceil32(cin)) bf16 slot order.  in_cm = 1: x (B,csplit,L) fp32 channel-major holds input channels
//     [0, csplit) and x2 (B,cin - csplit,L) the rest (csplit = cin, x2 = NULL: one tensor) -- the [xyz, feat] / [skip, interp]
//     concats of pointnet_utils.py:286-294, 318-321 are never built; needs l % 4 == 0.
//   out_mode = 0: y (B,L,ceil32(cout)) bf16 slot order; 1: y (B,cout,L) fp32; 2: y (B,cout) fp32 = max over the l <= 128 positions.
//   wimg packed with perm = 1 in every case (the LDS image is in slot order).  ab (in_cm = 0 only): (B,cin,2) or NULL; act
//   CAPTRA_ACT_NONE / RELU (any of the three for fp32 outputs); stats (out_mode 0 only) (B,T,cout,2) or NULL; bias_bs: 0, or cout
//   for a bias per cloud.  Returns -2 for shapes the kernel is not instantiated for: the caller uses captra_pointwise_mlp_bf16pm.
extern "C" int captra_dense_bf16_tile_ex(int b, int cin, int cout, long long l, int in_cm, const void *x, const float *x2, int csplit,
                                         const unsigned char *wimg, const float *bias_packed, long long bias_bs, const float *ab, int act,
                                         int out_mode, void *y, float *stats, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2 || out_mode < 0 || out_mode > 2) return -1;
    if (out_mode == 0 && act == ACT_SIGMOID_M05) return -1;
    if (in_cm && (ab != nullptr || csplit < 0 || csplit > cin || (csplit < cin && x2 == nullptr))) return -1;
    if (stats != nullptr && out_mode != 0) return -1;
    const int cp_in = (cin + 31) / 32 * 32, cp_out = (cout + 31) / 32 * 32;
    if (!in_cm && l * cp_in * 2 >= (1ll << 31)) return -2;
    if (in_cm && (l % 4 != 0 || l < 4)) return -2;
    if (out_mode == 2 && l > TB_P) return -2;
    if (cout < 64 && !(in_cm && cout >= 32)) return -2;
    if (b == 0 || l == 0) return 0;
    TbParams p;
    p.cin = cin; p.cout = cout; p.kst = (cin + 15) / 16; p.nt = (cout + 31) / 32; p.cp_in = cp_in; p.cp_out = cp_out; p.L = l;
    p.x = reinterpret_cast<const unsigned char *>(x); p.x2 = x2; p.csplit = in_cm ? csplit : cin;
    p.wimg = wimg; p.bias = bias_packed; p.bias_bs = bias_bs; p.ab = ab;
    p.y = reinterpret_cast<unsigned char *>(y); p.act = act; p.stats = stats; p.st_t = captra_dense_bf16_tile_stats_tiles(l);
    if (ab != nullptr && p.kst * 128 > 65536) return -2;
    hipStream_t s = (hipStream_t)stream;
    const bool aff = ab != nullptr, st = stats != nullptr;
    // few position tiles (the 128- / 512-point levels of the backbone): narrow row blocks, so that the launch has workgroups
    const long long ptiles = (long long)b * ((l + TB_P - 1) / TB_P);
    if (in_cm) {
#define TB_GO_CM(MT_, NW_)                                                                  \
    do {                                                                                    \
        if (out_mode == 0) return tb_launch<MT_, NW_, 1, false, 0, false>(b, p, s);         \
        if (out_mode == 1) return tb_launch<MT_, NW_, 1, false, 1, false>(b, p, s);         \
        return tb_launch<MT_, NW_, 1, false, 2, false>(b, p, s);                            \
    } while (0)
        if (p.nt >= 16 && ptiles >= 512) TB_GO_CM(2, 8);
        if (p.nt >= 8 && ptiles >= 128) TB_GO_CM(1, 8);
        TB_GO_CM(1, 4);
#undef TB_GO_CM
    }
#define TB_GO(MT_, NW_)                                                                     \
    do {                                                                                    \
        if (out_mode == 1) return tb_launch<MT_, NW_, 0, false, 1, false>(b, p, s);         \
        if (out_mode == 2) return tb_launch<MT_, NW_, 0, false, 2, false>(b, p, s);         \
        if (aff && st) return tb_launch<MT_, NW_, 0, true, 0, true>(b, p, s);               \
        if (aff) return tb_launch<MT_, NW_, 0, true, 0, false>(b, p, s);                    \
        if (st) return tb_launch<MT_, NW_, 0, false, 0, true>(b, p, s);                     \
        return tb_launch<MT_, NW_, 0, false, 0, false>(b, p, s);                            \
    } while (0)
    if (out_mode != 0 && aff) return -2;
    if (p.nt >= 16 && ptiles >= 512) TB_GO(2, 8);
    if (p.nt >= 5 && ptiles >= 256) TB_GO(2, 4);
    TB_GO(1, 4);
#undef TB_GO
}

extern "C" int captra_dense_bf16_tile(int b, int cin, int cout, long long l, const void *x, const unsigned char *wimg, const float *bias_packed,
                                      long long bias_bs, const float *ab, int act, void *y, float *stats, captra_stream_t stream) {
    return captra_dense_bf16_tile_ex(b, cin, cout, l, 0, x, nullptr, cin, wimg, bias_packed, bias_bs, ab, act, 0, y, stats, stream);
}

// Layers 1 + 2 of a Conv -> GroupNorm -> ReLU head in one launch (cin <= 128 -> 512 -> 512).  ab1 == NULL: the statistics
// pass -- stats (B,T,512,2) receives y1's partial sums (y1 = W1 x + b1), nothing else is written.  ab1 != NULL (from
// captra_gn_finalize_tm on those): y2 = W2 bf16(relu(a1 y1 + b1')) + b2 stored raw as (B,L,512) bf16 slot order, stats = y2's.
extern "C" int captra_head12_bf16_ex(int b, int cin, long long l, const void *x, const unsigned char *w1img, const float *bias1_packed,
                                  const float *ab1, const unsigned char *w2img, const float *bias2_packed, void *y2, float *stats,
                                  const captra_launch_opts *opts, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cin > 128 || l < 0 || stats == nullptr) return -1;
    if (ab1 != nullptr && (w2img == nullptr || y2 == nullptr || bias2_packed == nullptr)) return -1;
    if (l * 512 * 2 >= (1ll << 31)) return -2;
    if (b == 0 || l == 0) return 0;
    HbParams p;
    p.cin = cin; p.kst1 = (cin + 15) / 16; p.cp_x = (cin + 31) / 32 * 32; p.L = l;
    p.x = reinterpret_cast<const unsigned char *>(x); p.w1 = w1img; p.bias1 = bias1_packed; p.ab1 = ab1; p.w2 = w2img; p.bias2 = bias2_packed;
    p.y2 = reinterpret_cast<unsigned char *>(y2); p.stats = stats; p.st_t = captra_dense_bf16_tile_stats_tiles(l);
    p.dbg = g_tb_dbg;
    dim3 grid((unsigned)((l + TB_P - 1) / TB_P), b);
    hipStream_t s = (hipStream_t)stream;
    if (ab1 == nullptr) {
        CAPTRA_LAUNCH("pointwise_mlp", tb_head12_kernel<0>, grid, dim3(512), 32768, s, p);
        return captra_last_error();
    }
    static CaptraDeviceOnce once;
    static std::atomic<int> cus_of[128];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&tb_head12_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&tb_head12p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 8192);
        hipDeviceProp_t prop;
        cus_of[dev & 127].store((hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256);
        once.done();
    }
    const int cus_dev = cus_of[dev & 127].load() > 0 ? cus_of[dev & 127].load() : 256;
    const int cus = cus_dev - captra_reserved_cus(opts) > 0 ? cus_dev - captra_reserved_cus(opts) : 1;      // (captra_launch_opts::reserved_cus)
    const long long tpc = (l + TB_P - 1) / TB_P, ntiles = (long long)b * tpc;
    if (g_tb_persist && ntiles > cus && ntiles < (1ll << 30)) {
        // persistent: one workgroup per CU, contiguous runs of tiles
        HpParams pp;
        pp.q = p; pp.tpc = (int)tpc; pp.ntiles = (int)ntiles;
        pp.tpw = (int)((ntiles + cus - 1) / cus);
        const int nwg = (int)((ntiles + pp.tpw - 1) / pp.tpw);
        CAPTRA_LAUNCH("pointwise_mlp", tb_head12p_kernel, dim3(nwg), dim3(512), 131072 + 8192, s, pp);
        return captra_last_error();
    }
    CAPTRA_LAUNCH("pointwise_mlp", tb_head12_kernel<1>, grid, dim3(512), 131072, s, p);
    return captra_last_error();
}
extern "C" int captra_head12_bf16(int b, int cin, long long l, const void *x, const unsigned char *w1img, const float *bias1_packed,
                                  const float *ab1, const unsigned char *w2img, const float *bias2_packed, void *y2, float *stats,
                                  captra_stream_t stream) {
    return captra_head12_bf16_ex(b, cin, l, x, w1img, bias1_packed, ab1, w2img, bias2_packed, y2, stats, nullptr, stream);
}
