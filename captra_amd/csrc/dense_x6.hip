// f32x6 dense layer for gfx950: y = act(W x' + b) over (B, cin, L) -> (B, cout, L) with every product on
// v_mfma_f32_32x32x16_bf16 as six bf16 products of three-way splits (arithmetic contract: include/captra_hip.h "f32x6", csrc/sa_x6.hip),
// for the Conv1d -> GroupNorm -> ReLU chains of the rotation heads (reference network/models/blocks.py:150-165, 168-193):
// x' = relu(a x + b) with the previous layer's per-(cloud, channel) GroupNorm coefficients applied while the operand is staged,
// (sum, sum of squares) of the raw output per channel and 128 positions from the epilogue -- the interface of
// captra_pointwise_mlp_gn (csrc/pointwise_mlp.hip), whose exact-fp32 kernels this replaces in the opt-in mode.
//
// Structure: a 128-position x 256-channel tile per workgroup of FOUR waves, a wave = 128 positions x 64 channels (128 accumulator
// registers), TWO workgroups per CU: one workgroup's operand split (70 VALU per wave and k-step), barriers, prologue and store
// epilogue run under the other's MFMAs -- the first form, one 256 x 256 workgroup of eight waves per CU, left the matrix pipe 54 % busy
// (profiles/r06a_bench_f32x6_pmc_summary.txt: twice as many wait as active cycles).  k-steps of 16 channels:
//   positions every lane loads 8 channels of ONE position (dword loads, a wave = 2 rows x 128 B per instruction), applies the
//             GroupNorm coefficients + ReLU, splits, and writes three 16-byte fragment slots of a two-stage LDS buffer: the fragment
//             image is lane-linear both ways, no transposition anywhere.  The split of k-step kk + 1 runs in four pieces BETWEEN the
//             MFMA groups of k-step kk (pinned with masked sched_barriers: the scheduler otherwise sinks all of it behind the last MFMA);
//   weights   a wave's own six fragments (2 channel tiles x 3 parts, from the split image of captra_pack_dense_x6) straight into
//             REGISTERS, two sets: a k-step runs its two channel tiles one after the other and re-loads each tile's three fragments
//             with k-step kk + 2's as soon as its 24 MFMAs have issued.  (Until late round 6 they came through LDS by LDS-DMA, two
//             stages: a k-step's pieces then had ONE k-step of MFMAs to arrive in, waited for in front of the barrier -- 3-10 % slower
//             per launch, tools/ab_round6.sh; that form remains for cin % 32 != 0 and as CAPTRA_DX_WREG=0.)
// The MFMAs are FLIPPED (positions = rows): a lane owns a channel, so the statistics are sums over its own registers; the output
// tile leaves through LDS as whole 512-byte rows.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int dx_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned dx_pack(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float dx_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float dx_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void dx_split2(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
    p0 = dx_pack(a, b);
    const float ra = a - dx_lo(p0), rb = b - dx_hi(p0);
    p1 = dx_pack(ra, rb);
    p2 = dx_pack(ra - dx_lo(p1), rb - dx_hi(p1));
}
__device__ __forceinline__ f32x16 dx_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int DX_TP = 128, DX_TC = 256;                // positions / channels per workgroup
constexpr int DX_ASTAGE = 24 * 1024, DX_BSTAGE = 12 * 1024, DX_STAGE = DX_ASTAGE + DX_BSTAGE;
constexpr int DX_LDS = 2 * DX_STAGE;                   // + the coefficient table (cin x 8 bytes) behind it

// ---- weight image: [channel block of 256][k-step][channel tile 0..7][part 0..2] fragments of 1 KiB; fragment lane l = channel
// 256 cb + 32 t + (l & 31), k-slots 8 (l >> 5) .. + 7 = input channels 16 kk + 8 (l >> 5) + e (natural order) -------------------
__global__ void pack_dense_x6_kernel(int cin, int cout, int ldw, const float *__restrict__ wt, unsigned char *__restrict__ img) {
    const int kst = (cin + 15) / 16, ncb = (cout + DX_TC - 1) / DX_TC;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)ncb * kst * 24 * 512) return;
    const int el = (int)e & 7, lane = (int)(e >> 3) & 63;
    const long long f = e >> 9;
    const int part = (int)(f % 3), t = (int)((f / 3) % 8), kk = (int)((f / 24) % kst), cb = (int)(f / (24ll * kst));
    const int ch = DX_TC * cb + 32 * t + (lane & 31), k = 16 * kk + 8 * (lane >> 5) + el;
    const float v = (ch < cout && k < cin) ? wt[(size_t)k * ldw + ch] : 0.f;
    const __bf16 h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const __bf16 h2 = (__bf16)(r1 - (float)h1);
    const __bf16 h = part == 0 ? h0 : (part == 1 ? h1 : h2);
    reinterpret_cast<unsigned short *>(img)[e] = __builtin_bit_cast(unsigned short, h);
}

struct DxParams {
    int b, cin, cout, l;
    const float *x;             // (B,cin,L)
    const unsigned char *wimg;  // captra_pack_dense_x6
    const float *bias;          // packed bias (cout, zero padded)
    const float *ab;            // (B,cin,2) or null
    int act;
    float *y;                   // (B,cout,L)
    float *stats;               // (B,cout,T,2), T = L / 128, or null
    int npt, ncb;               // position tiles per cloud, channel blocks
};

// timing ablations (-DCAPTRA_DX_ABL=n, results wrong): 1 no wait for the k-step's loads, 2 no weight DMA after start-up, 3 no position
// loads after start-up, 4 neither wait nor barrier; register-weight form: 5 MFMAs + fragment reads + barrier only (no loads, no
// split after start-up), 6 the same without the barrier, 8 as 5 with the weight loads
#ifdef CAPTRA_DX_ABL
#define DX_ABL CAPTRA_DX_ABL
#else
#define DX_ABL 0
#endif
#define DX_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)

template <bool GN_IN, bool STATS, bool WREG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dense_x6_kernel(DxParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int wm = wave;                                // this wave's 64-channel block of the tile (all of its 128 positions)
    // workgroup -> (cloud, position tile, channel block): the channel blocks of one position tile are 8 workgroups apart, i.e. on
    // the same XCD (workgroups go round the eight XCDs), so the second reading of a position tile's x is an L2 hit
    const int per_cloud = p.npt * p.ncb;
    const int bq = blockIdx.x / per_cloud, r = blockIdx.x - bq * per_cloud;
    int pt, cb;
    if (p.npt % 8 == 0) {
        const int g = r / (8 * p.ncb), q = r - g * 8 * p.ncb;
        cb = q >> 3; pt = g * 8 + (q & 7);
    } else {
        pt = r / p.ncb; cb = r - pt * p.ncb;
    }
    const int kst = p.cin >> 4;
    const int p0 = pt * DX_TP;

    const unsigned long long img_addr = reinterpret_cast<unsigned long long>(p.wimg) + (size_t)cb * kst * DX_ASTAGE;
    const dx_i32x4 wsrc = {(int)(unsigned)img_addr, (int)(unsigned)(img_addr >> 32), kst * DX_ASTAGE, 0x00020000};
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned voff16 = lane * 16;
    // LDS-DMA of k-step kk's weight fragments into stage st: this wave's six pieces (asm: see csrc/sa_x6.hip)
    auto issue_w = [&](int kk, int st) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned dst = lds0 + st * DX_STAGE + (wave * 6 + i) * 1024;
            const unsigned soff = kk * DX_ASTAGE + (wave * 6 + i) * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(voff16), "s"(wsrc), "s"(soff) : "memory");
        }
    };
    // WREG: this wave's six weight fragments of a k-step go straight into registers -- nobody else reads them, and through LDS (two
    // stages) a k-step's pieces had ONE k-step of MFMAs to arrive in: the kernel ran at the L2 round trip.  Two register sets: a k-step
    // runs its row tiles one after the other, and each half's three fragments are re-loaded with k-step kk + 2's as soon as its 24
    // MFMAs have issued (1.5 k-steps ahead of their use)
    u32x4 wr[WREG ? 2 : 1][2][3];
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.wimg + (size_t)cb * kst * DX_ASTAGE), 0, kst * DX_ASTAGE, 0x00020000);
    auto load_w = [&](auto setc, auto tmc, int kk) {
        constexpr int SET = decltype(setc)::value, TM = decltype(tmc)::value;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
            wr[WREG ? SET : 0][TM][s3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)voff16, kk * DX_ASTAGE + ((2 * wm + TM) * 3 + s3) * 1024, 0));
    };
    // this lane's eight channels 16 kk + 8 h + (0..7) of position p0 + 32 wave + col
    const float *xp = p.x + ((size_t)bq * p.cin + 8 * h) * p.l + p0 + 32 * wave + col;
    auto load_x = [&](int kk, float (&xr)[8]) {
        const float *q = xp + (size_t)16 * kk * p.l;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = q[(size_t)i * p.l];
    };
    float *abt = reinterpret_cast<float *>(smem + DX_LDS);
    // GroupNorm coefficients + ReLU, split, three 16-byte fragment slots of position tile `wave`
    auto stage_x = [&](int kk, int st, const float (&xr)[8]) {
        float v[8];
        if constexpr (GN_IN) {
            const float4 *ap = reinterpret_cast<const float4 *>(abt + 2 * (16 * kk + 8 * h));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 c = ap[i];                 // (a, b) of two channels
                v[2 * i] = relu_bits(__builtin_fmaf(c.x, xr[2 * i], c.y));        // (the exact kernels' form: csrc/pointwise_mlp.hip)
                v[2 * i + 1] = relu_bits(__builtin_fmaf(c.z, xr[2 * i + 1], c.w));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = xr[i];
        }
        u32x4 f0, f1, f2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned a0, a1, a2;
            dx_split2(v[2 * i], v[2 * i + 1], a0, a1, a2);
            f0[i] = a0; f1[i] = a1; f2[i] = a2;
        }
        unsigned char *bp = smem + st * DX_STAGE + DX_ASTAGE + wave * 3072 + lane * 16;
        *reinterpret_cast<u32x4 *>(bp) = f0;
        *reinterpret_cast<u32x4 *>(bp + 1024) = f1;
        *reinterpret_cast<u32x4 *>(bp + 2048) = f2;
    };

    // WREG: the same in four pieces (two channels each) that a k-step places between its MFMA groups -- staged in one piece at the top
    // of the k-step, the split was a phase of ~85 instructions no MFMA of this wave covered
    u32x4 sf0, sf1, sf2;
    auto stage_piece = [&](auto ic, int kk, int st, const float (&xs)[8]) {
        constexpr int i = decltype(ic)::value;
        float v0 = xs[2 * i], v1 = xs[2 * i + 1];
        if constexpr (GN_IN) {
            const float4 c = reinterpret_cast<const float4 *>(abt + 2 * (16 * kk + 8 * h))[i];
            v0 = relu_bits(__builtin_fmaf(c.x, v0, c.y));
            v1 = relu_bits(__builtin_fmaf(c.z, v1, c.w));
        }
        unsigned a0, a1, a2;
        dx_split2(v0, v1, a0, a1, a2);
        sf0[i] = a0; sf1[i] = a1; sf2[i] = a2;
        if constexpr (i == 3) {
            unsigned char *bp = smem + st * DX_STAGE + DX_ASTAGE + wave * 3072 + lane * 16;
            *reinterpret_cast<u32x4 *>(bp) = sf0;
            *reinterpret_cast<u32x4 *>(bp + 1024) = sf1;
            *reinterpret_cast<u32x4 *>(bp + 2048) = sf2;
        }
    };

    // ---- prologue -------------------------------------------------------------------------------------------------------------
    if constexpr (GN_IN) {
        const float *src = p.ab + (size_t)bq * p.cin * 2;
        for (int e = tid; e < p.cin * 2; e += 256) abt[e] = src[e];
    }
    float xr[8];
    float xq[WREG ? 8 : 1];                              // WREG: the second position set (k-steps of odd parity)
    if constexpr (WREG) {
        load_x(0, xr);
        load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        load_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 0);
    } else {
        issue_w(0, 0);
        load_x(0, xr);
    }
    if constexpr (GN_IN) __syncthreads();
    stage_x(0, 0, xr);
    if constexpr (WREG) load_x(1, xq);
    else if (kst > 1) load_x(1, xr);
    {
        if constexpr (WREG) {
            load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, 1);
            load_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, 1);
        }
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[tm][tn][rr] = 0.f;
    DX_WAIT_VM(0);
    __syncthreads();

    // ---- k-steps ---------------------------------------------------------------------------------------------------------------
    auto kstep = [&](auto setc, int kk) {
        constexpr int SET = decltype(setc)::value;
        const int st = kk & 1;
        // the next k-step's operands: split what was loaded during the last phase, then ask for the k-step after it
        if constexpr (WREG) {
            // straight-line on purpose (clamped k-steps instead of branches: what the last k-steps stage and load is never used): at a
            // join the compiler's vmcnt bookkeeping assumes the path with the fewest loads in flight and waited for ALL of them.
            // Positions before weights: loads return in order, and the wait for the positions at the next split leaves the weights in flight
            // Two position sets: k-step kk asks for k-step kk + 2's at its top (into the set k-step kk - 1 has used up) and splits
            // k-step kk + 1's, loaded a k-step ago, between its MFMA groups.
#if DX_ABL < 5
            if constexpr (SET == 0) load_x(kk + 2 < kst ? kk + 2 : kst - 1, xr);
            else load_x(kk + 2 < kst ? kk + 2 : kst - 1, xq);
#endif
        } else if (kk + 1 < kst) {
            stage_x(kk + 1, st ^ 1, xr);
            {
#if DX_ABL != 3
                load_x(kk + 2 < kst ? kk + 2 : kst - 1, xr);        // (always issued, clamped: the load counts stay static)
#endif
#if DX_ABL != 2
                issue_w(kk + 1, st ^ 1);
#endif
            }
        }
        const unsigned char *ab_ = smem + st * DX_STAGE + lane * 16;
        if constexpr (WREG) {
            auto half = [&](auto tmc) {
                constexpr int TM = decltype(tmc)::value;
#pragma unroll
                for (int tn = 0; tn < 4; tn += 2) {
                    u32x4 xa[3], xb[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        xa[s] = *reinterpret_cast<const u32x4 *>(ab_ + DX_ASTAGE + (tn * 3 + s) * 1024);
                        xb[s] = *reinterpret_cast<const u32x4 *>(ab_ + DX_ASTAGE + ((tn + 1) * 3 + s) * 1024);
                    }
                    // two position tiles' chains issued alternately (see below)
                    f32x16 a = acc[TM][tn], b = acc[TM][tn + 1];
                    a = dx_mfma(xa[0], wr[SET][TM][2], a); b = dx_mfma(xb[0], wr[SET][TM][2], b);
                    a = dx_mfma(xa[2], wr[SET][TM][0], a); b = dx_mfma(xb[2], wr[SET][TM][0], b);
                    a = dx_mfma(xa[1], wr[SET][TM][1], a); b = dx_mfma(xb[1], wr[SET][TM][1], b);
                    a = dx_mfma(xa[0], wr[SET][TM][1], a); b = dx_mfma(xb[0], wr[SET][TM][1], b);
                    a = dx_mfma(xa[1], wr[SET][TM][0], a); b = dx_mfma(xb[1], wr[SET][TM][0], b);
                    a = dx_mfma(xa[0], wr[SET][TM][0], a); b = dx_mfma(xb[0], wr[SET][TM][0], b);
                    acc[TM][tn] = a; acc[TM][tn + 1] = b;
                    const int kn = kk + 1 < kst ? kk + 1 : kst - 1;
                    // (the piece stays between the two MFMA groups: fragment reads, loads and scalar code may cross the fences, VALU, LDS
                    // writes and MFMAs may not -- left alone the scheduler sinks the whole split behind the k-step's last MFMA)
#if DX_ABL < 5
                    __builtin_amdgcn_sched_barrier(0x124);
                    if (tn == 0) {
                        if constexpr (SET == 0) stage_piece(std::integral_constant<int, 2 * TM>{}, kn, st ^ 1, xq);
                        else stage_piece(std::integral_constant<int, 2 * TM>{}, kn, st ^ 1, xr);
                    } else {
                        if constexpr (SET == 0) stage_piece(std::integral_constant<int, 2 * TM + 1>{}, kn, st ^ 1, xq);
                        else stage_piece(std::integral_constant<int, 2 * TM + 1>{}, kn, st ^ 1, xr);
                    }
                    __builtin_amdgcn_sched_barrier(0x124);
#endif
                }
#if DX_ABL < 5 || DX_ABL == 8
                load_w(setc, tmc, kk + 2 < kst ? kk + 2 : kst - 1);
#endif
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
        } else {
        u32x4 wf[2][3];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int s = 0; s < 3; ++s) wf[tm][s] = *reinterpret_cast<const u32x4 *>(ab_ + ((2 * wm + tm) * 3 + s) * 1024);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            u32x4 xf[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) xf[s] = *reinterpret_cast<const u32x4 *>(ab_ + DX_ASTAGE + (tn * 3 + s) * 1024);
            {
                // the two row tiles' chains issued ALTERNATELY: back-to-back MFMAs on one accumulator run at the issue rate only while
                // nothing sits between them (a fragment read or a staging VALU in such a chain costs ~43 cycles, between MFMAs on
                // different accumulators ~6: MI355X_MICROARCH.md)
                f32x16 a = acc[0][tn], b = acc[1][tn];
                a = dx_mfma(xf[0], wf[0][2], a); b = dx_mfma(xf[0], wf[1][2], b);
                a = dx_mfma(xf[2], wf[0][0], a); b = dx_mfma(xf[2], wf[1][0], b);
                a = dx_mfma(xf[1], wf[0][1], a); b = dx_mfma(xf[1], wf[1][1], b);
                a = dx_mfma(xf[0], wf[0][1], a); b = dx_mfma(xf[0], wf[1][1], b);
                a = dx_mfma(xf[1], wf[0][0], a); b = dx_mfma(xf[1], wf[1][0], b);
                a = dx_mfma(xf[0], wf[0][0], a); b = dx_mfma(xf[0], wf[1][0], b);
                acc[0][tn] = a; acc[1][tn] = b;
            }
        }
        }
        if constexpr (!WREG) {
#if DX_ABL != 1 && DX_ABL != 4
            DX_WAIT_VM(0);                              // this wave's weight pieces of the next k-step have landed ...
#endif
        }
#if DX_ABL != 4 && DX_ABL != 6
        __syncthreads();                                // ... everybody's have, and everybody is done with this stage
#endif
    };
    if constexpr (WREG) {
#pragma unroll 1
        for (int kk = 0; kk < kst; kk += 2) {          // (kst even: the launcher's rule for this form)
            kstep(std::integral_constant<int, 0>{}, kk);
            kstep(std::integral_constant<int, 1>{}, kk + 1);
        }
    } else {
#pragma unroll 1
        for (int kk = 0; kk < kst; ++kk) kstep(std::integral_constant<int, 0>{}, kk);
    }

    // ---- epilogue: bias, activation, statistics of the raw output (a lane owns a channel: sums over its own registers) ------------
    // The tile leaves through LDS so that a store instruction writes whole rows -- 32 lanes x 16 bytes = the 512 contiguous bytes of a
    // channel's 128 positions, two channels per instruction (straight from the accumulators a lane's float4 is 4 of ITS channel's
    // positions, one instruction = 64 pieces of 16 bytes in 32 rows: 3 % slower per launch, tools/ab_round6.sh).  The wave's 32-channel
    // half tile goes to ITS OWN 16.5 KiB of the (now idle) operand stages, rows padded to 132 words: no barrier.
    constexpr int TR_ROW = 132;
    float *trw = reinterpret_cast<float *>(smem) + wave * (32 * TR_ROW);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int ch = DX_TC * cb + 64 * wm + 32 * tm + col;
        const float bias = p.bias[ch];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                v.x = acc[tm][tn][4 * q + 0] + bias; v.y = acc[tm][tn][4 * q + 1] + bias;
                v.z = acc[tm][tn][4 * q + 2] + bias; v.w = acc[tm][tn][4 * q + 3] + bias;
                if constexpr (STATS) {
                    s1 += (v.x + v.y) + (v.z + v.w);
                    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
                v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                *reinterpret_cast<float4 *>(trw + col * TR_ROW + 32 * tn + 8 * q + 4 * h) = v;
            }
        {
            // (same wave wrote it: LDS operations of a wave complete in order, the compiler's lgkmcnt wait covers the read-after-write)
            float *yr = p.y + ((size_t)bq * p.cout + DX_TC * cb + 64 * wm + 32 * tm + h) * p.l + p0 + 4 * col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 v = *reinterpret_cast<const float4 *>(trw + (2 * r + h) * TR_ROW + 4 * col);
                *reinterpret_cast<float4 *>(yr + (size_t)(2 * r) * p.l) = v;
            }
        }
        if constexpr (STATS) {
            const auto w1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
            const auto w2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
            if (h == 0) {
                const int T = p.l / 128;
                float2 o;
                o.x = __uint_as_float(w1[0]) + __uint_as_float(w1[1]);
                o.y = __uint_as_float(w2[0]) + __uint_as_float(w2[1]);
                *reinterpret_cast<float2 *>(p.stats + (((size_t)bq * p.cout + ch) * T + p0 / 128) * 2) = o;
            }
        }
    }
}

}  // namespace

extern "C" long long captra_dense_x6_image_bytes(int cin, int cout) {
    if (cin < 1 || cout < 1) return -1;
    return (long long)((cout + DX_TC - 1) / DX_TC) * ((cin + 15) / 16) * DX_ASTAGE;
}

// wt_packed: the layer's PACKED fp32 buffer (row-major part W'^T (cin, pad128(cout)))
extern "C" int captra_pack_dense_x6(int cin, int cout, const float *wt_packed, unsigned char *img, captra_stream_t stream) {
    if (cin < 1 || cout < 1 || wt_packed == nullptr || img == nullptr) return -1;
    const long long n = captra_dense_x6_image_bytes(cin, cout) / 2;
    CAPTRA_LAUNCH("pack_weights", pack_dense_x6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cin, cout,
                  (cout + 127) / 128 * 128, wt_packed, img);
    return captra_last_error();
}

// the stats_t captra_pointwise_mlp_x6 writes: one (sum, sum of squares) per channel and 128 positions
extern "C" int captra_pointwise_mlp_x6_tiles(long long l) { return (int)(l / 128); }

// As captra_pointwise_mlp_gn in the f32x6 arithmetic.  -2: shape outside the kernel (cin % 16, cout % 256, l % 128, cin > 1024).
extern "C" int captra_pointwise_mlp_x6(int b, int cin, int cout, long long l, const float *x, const unsigned char *wimg, const float *bias_packed,
                                       const float *ab_in, int act, float *y, float *stats_out, int stats_t, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || x == nullptr || wimg == nullptr || bias_packed == nullptr || y == nullptr) return -1;
    if (cin % 16 != 0 || cout % DX_TC != 0 || l % DX_TP != 0 || cin > 1024 || l >= (1ll << 30)) return -2;
    if (stats_out != nullptr && (act != ACT_NONE || stats_t != (int)(l / 128))) return -1;
    if (b == 0 || l == 0) return 0;
    DxParams p;
    p.b = b; p.cin = cin; p.cout = cout; p.l = (int)l; p.x = x; p.wimg = wimg; p.bias = bias_packed; p.ab = ab_in; p.act = act; p.y = y;
    p.stats = stats_out; p.npt = (int)(l / DX_TP); p.ncb = cout / DX_TC;
    const long long grid = (long long)b * p.npt * p.ncb;
    if (grid >= (1ll << 31)) return -2;
    const int lds = DX_LDS + cin * 8;
    static const bool dx_wreg = [] { const char *e = getenv("CAPTRA_DX_WREG"); return e == nullptr || e[0] != '0'; }();
#define DX_LAUNCH(GN_, ST_)                                                                                             \
    do {                                                                                                                \
        auto kern = (dx_wreg && cin % 32 == 0) ? dense_x6_kernel<GN_, ST_, true> : dense_x6_kernel<GN_, ST_, false>; \
        static CaptraDeviceOnce once;                                                                                   \
        if (once.first_use()) {                                                                                         \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DX_LDS + 1024 * 8) != hipSuccess) \
                return (int)hipGetLastError();                                                                          \
            once.done();                                                                                                \
        }                                                                                                               \
        CAPTRA_LAUNCH("pointwise_mlp_x6", kern, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);          \
    } while (0)
    if (ab_in != nullptr) { if (stats_out != nullptr) DX_LAUNCH(true, true); else DX_LAUNCH(true, false); }
    else { if (stats_out != nullptr) DX_LAUNCH(false, true); else DX_LAUNCH(false, false); }
#undef DX_LAUNCH
    return captra_last_error();
}
