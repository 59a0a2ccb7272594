// bf16-native dense layers for gfx950 (BASELINE.json configs[2]): the Conv1d 1x1 -> GroupNorm -> ReLU chains of the rotation
// heads (reference network/models/blocks.py:147-193) with the activations kept in HBM as bf16, POINT-major:
//
//   tensor (B, L, CP) bf16, CP = ceil32(C), channels in SLOT ORDER: inside every aligned block of 16 channels, memory slot s
//   holds channel perm[s] = {0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15} -- the order in which a 32x32 MFMA accumulator tile hands
//   its rows to a lane (csrc/sa_bf16.hip header), so a producer's epilogue is 8 v_cvt_pk_bf16_f32 (+ 8 v_pk_max_i16 for a
//   ReLU) and two 16-byte stores per tile, and a consumer's B operand is ONE 16-byte load per lane and k-step (the fp32
//   tensors of round 2 took eight dword loads + four conversions, and twice the bytes).
//
// Contract per layer: y = act(b + sum_k bf16(w[k]) * bf16(x[k])), fp32 accumulation; with `ab` the input is
// x = bf16(relu(a[c] * xraw + b[c])) -- the GroupNorm of the layer that produced xraw, applied while the operand is loaded,
// exactly as the fp32 path's pw_direct_kernel<..., AFF> does; xraw is that layer's output as stored (rounded to bf16), and
// the statistics are taken from the stored tensor (captra_gn_stats_bf16pm), so the normalisation is self-consistent.
//
// Kernel: no LDS for operands, no barriers; wave tile (TM*32 rows) x (TN*32 positions); A operands from the layer's FRAGMENT
// IMAGE (captra_pack_dense_bf16: frag (t,kk) = 64 lanes x 16 bytes, coalesced 1 KiB loads), three k-steps in flight.
#include "common.h"
#include "bf16_dense.h"

namespace {

// ---- weight image: [NT][KST][64 lanes][8] bf16; perm != 0: k-slot s of k-step kk = input channel 16kk + perm[s] ------------
__global__ void pack_dense_bf16_kernel(int cin, int cout, int ldw, int perm, const float *__restrict__ wt, unsigned short *__restrict__ img) {
    const int kst = (cin + 15) / 16, nt = (cout + 31) / 32;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nt * kst * 512) return;
    const int f = e >> 9, lane = (e >> 3) & 63, el = e & 7;
    const int t = f / kst, kk = f % kst;
    const int slot = 8 * (lane >> 5) + el;
    const int row = 32 * t + (lane & 31), k = 16 * kk + (perm ? db_perm(slot) : slot);
    const float v = (row < cout && k < cin) ? wt[(size_t)k * ldw + row] : 0.f;
    const __bf16 h = (__bf16)v;
    img[e] = __builtin_bit_cast(unsigned short, h);
}

struct DbParams {
    int cin, cout, kst, nt;
    int cp_in, cp_out;        // channel stride (elements) of a point-major input / output
    long long L;
    const void *x;            // IN_PM: (B,L,cp_in) bf16 slot order; else (B,cin,L) fp32
    const unsigned char *wimg;
    const float *bias;        // packed fp32 (ceil128(cout)); cloud b reads bias + b * bias_bs (0: one bias for all)
    long long bias_bs;
    const float *ab;          // AFF: (B,cin,2) fp32
    void *y;                  // OUT_PM: (B,L,cp_out) bf16 slot order; else (B,cout,L) fp32
    int act;
    float *stats;             // ST: (B,T,cout,2) partial (sum, sum of squares) of the STORED values, T = ceil(L / 64)
    int st_t;
};

template <int TM, int TN, bool IN_PM, bool AFF, bool OUT_PM, bool ST = false>
__global__ __launch_bounds__(256, 2) void pw_bf16pm_kernel(DbParams p) {
    static_assert(!AFF || IN_PM, "the on-load GroupNorm needs the point-major input");
    static_assert(!ST || (OUT_PM && TN == 2), "statistics of the stored bf16 tensor, chunks of 64 positions");
    constexpr int NS = 3;                                  // k-steps in flight
    extern __shared__ __attribute__((aligned(16))) float aff_tab[];   // AFF: [kst][2 halves][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int b = blockIdx.z;
    const int t0 = blockIdx.y * TM;
    const long long pos0 = ((long long)blockIdx.x * 4 + wave) * TN * 32;
    if constexpr (AFF) {
        const float *abp = p.ab + (size_t)b * p.cin * 2;
        for (int e = tid; e < p.kst * 32; e += 256) {
            const int kk = e >> 5, hh = (e >> 4) & 1, i = e & 15;
            const int c = 16 * kk + db_perm(8 * hh + (i & 7));
            aff_tab[e] = c < p.cin ? abp[2 * c + (i >> 3)] : 0.f;
        }
        __syncthreads();
    }
    if (pos0 >= p.L) return;                               // wave-uniform; no barrier below
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.wimg, 0, p.nt * p.kst * 1024, 0x00020000);
    __amdgpu_buffer_rsrc_t xsrc;
    int xvoff[TN];
    int woff[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int t = t0 + tm < p.nt ? t0 + tm : p.nt - 1;     // clamped row tile: computed, never stored
        woff[tm] = t * p.kst * 1024;
    }
    if constexpr (IN_PM) {
        const __bf16 *xb = reinterpret_cast<const __bf16 *>(p.x) + (size_t)b * p.L * p.cp_in;
        xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)(p.L * p.cp_in * 2), 0x00020000);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            long long c = pos0 + tn * 32 + col;
            if (c >= p.L) c = p.L - 1;                         // clamped column: computed, never stored
            xvoff[tn] = (int)((c * p.cp_in + 8 * h) * 2);
        }
    } else {
        const float *xb = reinterpret_cast<const float *>(p.x) + (size_t)b * p.cin * p.L;
        xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.cin * p.L * 4), 0x00020000);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            long long c = pos0 + tn * 32 + col;
            if (c >= p.L) c = p.L - 1;
            xvoff[tn] = (int)(((long long)(8 * h) * p.L + c) * 4);   // rows >= cin fall outside the buffer and read as 0
        }
    }
    const int xrow = (int)(p.L * 4);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const float *bp = p.bias + (size_t)b * p.bias_bs + (t0 + tm < p.nt ? t0 + tm : p.nt - 1) * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = bv;
        }
    }
    u32x4 A[NS][TM];
    u32x4 Bp[NS][TN];                 // IN_PM
    float Bf[IN_PM ? 1 : NS][IN_PM ? 1 : TN][8];
    const int kst = p.kst;
    auto load = [&](int s, int kk) {
        kk = kk < kst ? kk : kst - 1;                          // over-read of the last k-step (never multiplied)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, woff[tm] + kk * 1024, 0);
        if constexpr (IN_PM) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) Bp[s][tn] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, xvoff[tn], kk * 32, 0);
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    Bf[s][tn][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, xvoff[tn] + (kk * 16 + i) * xrow, 0, 0));
        }
    };
    auto mma = [&](int s, int kk) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            u32x4 bb;
            if constexpr (IN_PM) {
                bb = Bp[s][tn];
                if constexpr (AFF) bb = db_affine(bb, aff_tab + (kk * 2 + h) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) bb[i] = db_pack(Bf[s][tn][2 * i], Bf[s][tn][2 * i + 1]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = db_mfma(A[s][tm], bb, acc[tm][tn]);
        }
    };
    load(0, 0);
    load(1, 1);
    for (int kk = 0; kk < kst; kk += NS) {
        load(2, kk + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, kk);
        __builtin_amdgcn_sched_barrier(0);
        load(0, kk + 3);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < kst) mma(1, kk + 1);
        __builtin_amdgcn_sched_barrier(0);
        load(1, kk + 4);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 2 < kst) mma(2, kk + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue ------------------------------------------------------------------------------------------
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        if (t0 + tm >= p.nt) continue;
        float sv[ST ? 32 : 1];
        if constexpr (ST) {
#pragma unroll
            for (int i = 0; i < 32; ++i) sv[i] = 0.f;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const long long c = pos0 + tn * 32 + col;
            if (c >= p.L) continue;
            if constexpr (OUT_PM) {
                __bf16 *yp = reinterpret_cast<__bf16 *>(p.y) + ((size_t)b * p.L + c) * p.cp_out + 32 * (t0 + tm) + 8 * h;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    u32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = db_pack(acc[tm][tn][8 * jj + 2 * i], acc[tm][tn][8 * jj + 2 * i + 1]);
                        if (p.act == ACT_RELU) v[i] = db_relu2(v[i]);
                    }
                    *reinterpret_cast<u32x4 *>(yp + 16 * jj) = v;
                    if constexpr (ST) db_stats_acc(sv, jj, v);
                }
            } else {
                const int row0 = 32 * (t0 + tm) + 4 * h;
                float *yp = reinterpret_cast<float *>(p.y) + ((size_t)b * p.cout + row0) * p.L + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (row0 + ro < p.cout) yp[(size_t)ro * p.L] = apply_act(acc[tm][tn][r], p.act);
                }
            }
        }
        if constexpr (ST)
            db_stats_tile(sv, col, h, t0 + tm, p.cout, p.stats + (size_t)b * p.cout * p.st_t * 2, p.st_t, (int)(pos0 >> 6));
    }
}

// ---- the wide GroupNorm-consuming layers: one transform per operand element ------------------------------------------------
// pw_bf16pm_kernel<.., AFF> applies relu(a x + b) to its B fragments in every wave, i.e. once per 128-row block: for the
// 512-wide layers every input element is unpacked, scaled, rounded and re-packed FOUR times (28 VALU instructions per fragment
// beside 8 MFMAs: the kernel ran at 0.22 of the matrix pipe, 64 % of its wave cycles issue stalls).  Here the four waves of a
// workgroup own the four row blocks of the SAME 64 positions and share the operand: per super-step of four k-steps each wave
// loads and transforms two of the eight B fragments, parks them in LDS (double-buffered, one barrier per super-step), and all
// four read the eight fragments back with conflict-free ds_read_b128 -- 2.8 instructions per MFMA instead of 10.
// A operands: fragment image, three k-steps in flight, as in pw_bf16pm_kernel.
template <int TM, bool OUT_PM, bool ST = false>
__global__ __launch_bounds__(256, 2) void pw_bf16pm_affs_kernel(DbParams p) {
    static_assert(!ST || OUT_PM, "statistics of the stored bf16 tensor");
    constexpr int TN = 2, SS = 4;                          // position tiles per wave, k-steps per super-step
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *aff_tab = reinterpret_cast<float *>(lds);       // [kst][2 halves][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int b = blockIdx.z;
    const int t0 = (blockIdx.y * 4 + wave) * TM;           // this wave's first row tile
    const long long pos0 = (long long)blockIdx.x * TN * 32;
    const int kst = p.kst;
    const int nss = (kst + SS - 1) / SS;
    unsigned char *bbuf = lds + (size_t)((kst * 32 * 4 + 1023) / 1024) * 1024;     // 2 x SS*TN KiB of transformed B fragments
    {
        const float *abp = p.ab + (size_t)b * p.cin * 2;
        for (int e = tid; e < kst * 32; e += 256) {
            const int kk = e >> 5, hh = (e >> 4) & 1, i = e & 15;
            const int c = 16 * kk + db_perm(8 * hh + (i & 7));
            aff_tab[e] = c < p.cin ? abp[2 * c + (i >> 3)] : 0.f;
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.wimg, 0, p.nt * kst * 1024, 0x00020000);
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(p.x) + (size_t)b * p.L * p.cp_in;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)(p.L * p.cp_in * 2), 0x00020000);
    // the two fragments this wave prepares per super-step: position tile wave & 1, k-steps 2 (wave >> 1) + {0, 1}
    const int my_tn = wave & 1, my_k0 = 2 * (wave >> 1);
    long long mc = pos0 + my_tn * 32 + col;
    if (mc >= p.L) mc = p.L - 1;                           // clamped column: computed, never stored
    const int xvoff = (int)((mc * p.cp_in + 8 * h) * 2);
    int woff[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) woff[tm] = (t0 + tm < p.nt ? t0 + tm : p.nt - 1) * kst * 1024;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const float *bp = p.bias + (size_t)b * p.bias_bs + (t0 + tm < p.nt ? t0 + tm : p.nt - 1) * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = bv;
        }
    }
    constexpr int NS = SS;                                 // A register sets: k-step kk lives in set kk % SS, loaded three k-steps ahead
    u32x4 A[NS][TM];
    auto loadA = [&](int s, int kk) {
        kk = kk < kst ? kk : kst - 1;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, woff[tm] + kk * 1024, 0);
    };
    u32x4 raw0, raw1;                                      // this wave's two untransformed fragments of the NEXT super-step
    auto loadB = [&](int ss) {
        int k0 = ss * SS + my_k0, k1 = k0 + 1;             // (ss may run past the last super-step: clamped, loaded, never used)
        k0 = k0 < kst ? k0 : kst - 1;                      // beyond the last k-step: loaded and parked, never multiplied
        k1 = k1 < kst ? k1 : kst - 1;
        raw0 = __builtin_amdgcn_raw_buffer_load_b128(xsrc, xvoff, k0 * 32, 0);
        raw1 = __builtin_amdgcn_raw_buffer_load_b128(xsrc, xvoff, k1 * 32, 0);
    };
    // park this wave's two fragments of super-step `ss_` (transformed from raw0 / raw1) in buffer ss_ & 1
    auto park = [&](int ss_) {
        const int k0 = ss_ * SS + my_k0;
        unsigned char *dst = bbuf + (size_t)(ss_ & 1) * (SS * TN * 1024);
        const u32x4 f0 = db_affine(raw0, aff_tab + ((k0 < kst ? k0 : kst - 1) * 2 + h) * 16);
        const u32x4 f1 = db_affine(raw1, aff_tab + ((k0 + 1 < kst ? k0 + 1 : kst - 1) * 2 + h) * 16);
        *reinterpret_cast<u32x4 *>(dst + ((my_k0 + 0) * TN + my_tn) * 1024 + lane * 16) = f0;
        *reinterpret_cast<u32x4 *>(dst + ((my_k0 + 1) * TN + my_tn) * 1024 + lane * 16) = f1;
    };
    loadB(0);
    loadA(0, 0);
    loadA(1, 1);
    loadA(2, 2);
    park(0);
    loadB(1);                                              // (clamped k-steps: always issued, so the load counts stay static)
    __syncthreads();
    for (int ss = 0; ss < nss; ++ss) {
        // four k-steps on the shared fragments of buffer ss & 1; between the second and the third this wave transforms and
        // parks its two fragments of the NEXT super-step (the VALU work issues while the MFMAs of k-steps 0-1 execute) and
        // fetches the ones after; one barrier per super-step
        const unsigned char *src = bbuf + (size_t)(ss & 1) * (SS * TN * 1024);
#pragma unroll
        for (int ks = 0; ks < SS; ++ks) {
            const int kk = ss * SS + ks;
            loadA((ks + 3) % NS, kk + 3);                  // three k-steps (24 MFMAs, ~770 cycles) ahead: an L2 hit under load
            if (kk < kst) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const u32x4 bb = *reinterpret_cast<const u32x4 *>(src + (ks * TN + tn) * 1024 + lane * 16);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = db_mfma(A[ks % NS][tm], bb, acc[tm][tn]);
                }
            }
            if (ks == 1) {
                park(ss + 1);
                loadB(ss + 2);
            }
        }
        __syncthreads();
    }
    // ---- epilogue (as pw_bf16pm_kernel) -------------------------------------------------------------------------
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        if (t0 + tm >= p.nt) continue;
        float sv[ST ? 32 : 1];
        if constexpr (ST) {
#pragma unroll
            for (int i = 0; i < 32; ++i) sv[i] = 0.f;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const long long c = pos0 + tn * 32 + col;
            if (c >= p.L) continue;
            if constexpr (OUT_PM) {
                __bf16 *yp = reinterpret_cast<__bf16 *>(p.y) + ((size_t)b * p.L + c) * p.cp_out + 32 * (t0 + tm) + 8 * h;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    u32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = db_pack(acc[tm][tn][8 * jj + 2 * i], acc[tm][tn][8 * jj + 2 * i + 1]);
                        if (p.act == ACT_RELU) v[i] = db_relu2(v[i]);
                    }
                    *reinterpret_cast<u32x4 *>(yp + 16 * jj) = v;
                    if constexpr (ST) db_stats_acc(sv, jj, v);
                }
            } else {
                const int row0 = 32 * (t0 + tm) + 4 * h;
                float *yp = reinterpret_cast<float *>(p.y) + ((size_t)b * p.cout + row0) * p.L + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (row0 + ro < p.cout) yp[(size_t)ro * p.L] = apply_act(acc[tm][tn][r], p.act);
                }
            }
        }
        if constexpr (ST)
            db_stats_tile(sv, col, h, t0 + tm, p.cout, p.stats + (size_t)b * p.cout * p.st_t * 2, p.st_t, (int)(pos0 >> 6));
    }
}

// ---- GroupNorm partial statistics of a stored point-major tensor --------------------------------------------------------------
// x (B,L,CP) bf16 slot order -> stats (B,C,T,2) fp32: per channel and chunk of `pch` positions (sum, sum of squares) of the
// STORED values; captra_gn_finalize turns them into the (a, b) the consumer applies.  Block = (chunk, cloud); thread = one
// 16-byte channel group of one position per step; fixed summation order (no atomics).
__global__ __launch_bounds__(256) void gn_stats_bf16pm_kernel(int c, int cp, long long L, int pch, int T, const unsigned short *__restrict__ x,
                                                              float *__restrict__ stats) {
    __shared__ float red[256 * 16];
    const int tid = threadIdx.x;
    const int G = cp / 8, npar = 256 / G;
    const int g = tid % G, pl = tid / G;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const long long p0 = (long long)chunk * pch, p1 = p0 + pch < L ? p0 + pch : L;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    const uint4 *xb = reinterpret_cast<const uint4 *>(x + (size_t)b * L * cp);
    for (long long pos = p0 + pl; pos < p1; pos += npar) {
        const uint4 v = xb[pos * G + g];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
            s[2 * i] += lo; q[2 * i] = __builtin_fmaf(lo, lo, q[2 * i]);
            s[2 * i + 1] += hi; q[2 * i + 1] = __builtin_fmaf(hi, hi, q[2 * i + 1]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[tid * 16 + i] = s[i];
        red[tid * 16 + 8 + i] = q[i];
    }
    __syncthreads();
    // thread (g2, e) sums the npar position-lanes of channel group g2, element e
    for (int o = tid; o < G * 8; o += 256) {
        const int g2 = o >> 3, e = o & 7;
        float sm = 0.f, sq = 0.f;
        for (int k = 0; k < npar; ++k) {
            sm += red[(k * G + g2) * 16 + e];
            sq += red[(k * G + g2) * 16 + 8 + e];
        }
        const int ch = 16 * (g2 >> 1) + db_perm(8 * (g2 & 1) + e);
        if (ch < c) {
            float *dst = stats + (((size_t)b * c + ch) * T + chunk) * 2;
            dst[0] = sm;
            dst[1] = sq;
        }
    }
}

template <int TM, bool OUT_PM, bool ST = false>
int db_launch_affs(int b, const DbParams &p, hipStream_t s) {
    dim3 grid((unsigned)((p.L + 63) / 64), (p.nt + 4 * TM - 1) / (4 * TM), b);
    const int lds = (p.kst * 32 * 4 + 1023) / 1024 * 1024 + 2 * 4 * 2 * 1024;
    CAPTRA_LAUNCH("pointwise_mlp", (pw_bf16pm_affs_kernel<TM, OUT_PM, ST>), grid, dim3(256), lds, s, p);
    return captra_last_error();
}

template <int TM, int TN, bool IN_PM, bool AFF, bool OUT_PM, bool ST = false>
int db_launch(int b, const DbParams &p, hipStream_t s) {
    dim3 grid((unsigned)((p.L + 4 * TN * 32 - 1) / (4 * TN * 32)), (p.nt + TM - 1) / TM, b);
    const int lds = AFF ? p.kst * 32 * 4 : 0;
    CAPTRA_LAUNCH("pointwise_mlp", (pw_bf16pm_kernel<TM, TN, IN_PM, AFF, OUT_PM, ST>), grid, dim3(256), lds, s, p);
    return captra_last_error();
}

}  // namespace

static CAPTRA_KNOB int g_db_affs = 1;      // experiment knob: 0 = every wave transforms its own operand (pw_bf16pm_kernel<.., AFF>)
extern "C" void captra_dense_bf16_set_shared_affine(int on) { g_db_affs = on; }

extern "C" long long captra_dense_bf16_image_bytes(int cin, int cout) {
    if (cin < 1 || cout < 1) return -1;
    return (long long)((cout + 31) / 32) * ((cin + 15) / 16) * 1024;
}

// wt_packed: the layer's packed fp32 buffer (row-major part W'^T (ceil32(cin), ceil128(cout))).  perm = 1 for a layer whose
// input is a point-major slot-order tensor, 0 for a channel-major fp32 input.
extern "C" int captra_pack_dense_bf16(int cin, int cout, int perm, const float *wt_packed, unsigned char *img, captra_stream_t stream) {
    if (cin < 1 || cout < 1) return -1;
    const long long n = captra_dense_bf16_image_bytes(cin, cout) / 2;
    CAPTRA_LAUNCH("pack_weights", pack_dense_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cin, cout,
                  (cout + 127) / 128 * 128, perm, wt_packed, reinterpret_cast<unsigned short *>(img));
    return captra_last_error();
}

// One dense layer.  in_pm: x is (B,L,ceil32(cin)) bf16 slot order (image packed with perm = 1), else (B,cin,L) fp32 (perm = 0).
// ab (in_pm only): (B,cin,2) GroupNorm coefficients of the producing layer, applied as relu(a x + b) on load; or NULL.
// out_pm: y is (B,L,ceil32(cout)) bf16 slot order, else (B,cout,L) fp32.  act: CAPTRA_ACT_NONE / RELU (out_pm), any (fp32 out).
static int db_dispatch(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg, const float *bias_packed,
                       const float *ab, int act, int out_pm, void *y, float *stats, hipStream_t s, long long bias_bs = 0) {
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2) return -1;
    if (ab != nullptr && !in_pm) return -1;
    if (out_pm && act == ACT_SIGMOID_M05) return -1;
    if (stats != nullptr && !out_pm) return -1;
    const int cp_in = (cin + 31) / 32 * 32, cp_out = (cout + 31) / 32 * 32;
    if (in_pm ? l * cp_in * 2 >= (1ll << 31) : (long long)cin * l * 4 >= (1ll << 31)) return -2;
    if (b == 0 || l == 0) return 0;
    DbParams p;
    p.cin = cin; p.cout = cout; p.kst = (cin + 15) / 16; p.nt = (cout + 31) / 32; p.cp_in = cp_in; p.cp_out = cp_out; p.L = l;
    p.x = x; p.wimg = wimg; p.bias = bias_packed; p.ab = ab; p.y = y; p.act = act;
    p.stats = stats; p.st_t = (int)((l + 63) / 64);
    p.bias_bs = bias_bs;
    if (ab != nullptr && p.kst * 32 * 4 > 64 * 1024) return -2;
    const bool st = stats != nullptr;
    if (in_pm && ab != nullptr && p.nt >= 8 && g_db_affs) {
        // wide GroupNorm-consuming layers: the operand transformed once per workgroup and shared through LDS
        if (p.nt >= 16) return st ? db_launch_affs<4, true, true>(b, p, s) : out_pm ? db_launch_affs<4, true>(b, p, s) : db_launch_affs<4, false>(b, p, s);
        return st ? db_launch_affs<2, true, true>(b, p, s) : out_pm ? db_launch_affs<2, true>(b, p, s) : db_launch_affs<2, false>(b, p, s);
    }
    const bool wide = p.nt >= 4;
#define DB_GO(TM_, TN_)                                                                              \
    do {                                                                                             \
        if (st && in_pm && ab) return db_launch<TM_, TN_, true, true, true, true>(b, p, s);          \
        if (st && in_pm) return db_launch<TM_, TN_, true, false, true, true>(b, p, s);               \
        if (st) return db_launch<TM_, TN_, false, false, true, true>(b, p, s);                       \
        if (in_pm && ab && out_pm) return db_launch<TM_, TN_, true, true, true>(b, p, s);            \
        if (in_pm && ab) return db_launch<TM_, TN_, true, true, false>(b, p, s);                     \
        if (in_pm && out_pm) return db_launch<TM_, TN_, true, false, true>(b, p, s);                 \
        if (in_pm) return db_launch<TM_, TN_, true, false, false>(b, p, s);                          \
        if (out_pm) return db_launch<TM_, TN_, false, false, true>(b, p, s);                         \
        return db_launch<TM_, TN_, false, false, false>(b, p, s);                                    \
    } while (0)
    if (wide) DB_GO(4, 2);
    if (p.nt >= 2) DB_GO(2, 2);
    DB_GO(1, 2);
#undef DB_GO
}

extern "C" int captra_pointwise_mlp_bf16pm(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg,
                                           const float *bias_packed, const float *ab, int act, int out_pm, void *y, captra_stream_t stream) {
    return db_dispatch(b, cin, cout, l, in_pm, x, wimg, bias_packed, ab, act, out_pm, y, nullptr, (hipStream_t)stream);
}

// The layer with a bias PER CLOUD: bias (B, cout) fp32, cout a multiple of 32 (a feature-propagation layer whose second input
// is one vector per cloud -- pointnet_utils.py:265-268, S == 1: W [x; v 1^T] + b = W1 x + (W2 v + b) -- takes the bracket as its bias).
extern "C" int captra_pointwise_mlp_bf16pm_cb(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg,
                                              const float *bias_per_cloud, int act, int out_pm, void *y, captra_stream_t stream) {
    if (cout % 32 != 0) return -1;
    return db_dispatch(b, cin, cout, l, in_pm, x, wimg, bias_per_cloud, nullptr, act, out_pm, y, nullptr, (hipStream_t)stream, cout);
}

// The same layer with a point-major bf16 output, also leaving the GroupNorm partial statistics of what it stored:
// stats (B,T,cout,2) fp32 TILE-major, T = captra_dense_bf16_stats_tiles(l) (chunks of 64 positions), for captra_gn_finalize_tm.
extern "C" int captra_dense_bf16_stats_tiles(long long l) { return (int)((l + 63) / 64); }
extern "C" int captra_pointwise_mlp_bf16pm_stats(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg,
                                                 const float *bias_packed, const float *ab, int act, void *y, float *stats,
                                                 captra_stream_t stream) {
    if (stats == nullptr) return -1;
    return db_dispatch(b, cin, cout, l, in_pm, x, wimg, bias_packed, ab, act, 1, y, stats, (hipStream_t)stream);
}

extern "C" int captra_gn_stats_bf16pm_tiles(long long l) { return (int)((l + 127) / 128); }

// x (B,L,ceil32(c)) bf16 slot order -> stats (B,c,T,2), T = captra_gn_stats_bf16pm_tiles(l) (chunks of 128 positions)
extern "C" int captra_gn_stats_bf16pm(int b, int c, long long l, const void *x, float *stats, captra_stream_t stream) {
    if (b < 0 || c < 1 || l < 1) return -1;
    const int cp = (c + 31) / 32 * 32, G = cp / 8;
    if (256 % G != 0) return -2;
    if (b == 0) return 0;
    const int T = captra_gn_stats_bf16pm_tiles(l);
    CAPTRA_LAUNCH("gn_stats", gn_stats_bf16pm_kernel, dim3(T, b), dim3(256), 0, (hipStream_t)stream, c, cp, l, 128, T,
                  reinterpret_cast<const unsigned short *>(x), stats);
    return captra_last_error();
}
