// f32x6 set-abstraction scale for gfx950: the body of the loop over radii of PointNetSetAbstractionMsg.forward (reference
// network/models/pointnet_utils.py:228-248: gather, centre subtraction, cat, 3 x (Conv2d 1x1 + BN + ReLU), max over K) with every
// product of the 128..256-wide layers evaluated on v_mfma_f32_32x32x16_bf16 as SIX bf16 products of a THREE-WAY split of both
// operands, fp32 accumulation -- the opt-in arithmetic `mlp_dtype = "f32x6"`.
//
// Arithmetic contract (include/captra_hip.h "f32x6"): an fp32 number v is held as v = v0 + v1 + v2 EXACTLY, v0 = bf16(v),
// v1 = bf16(v - v0), v2 = bf16(v - v0 - v1) (round to nearest even; the two residuals are exact in fp32 and the third part has at most
// eight significant bits).  A layer is y = act(b + sum_k [w0 x0 + w0 x1 + w1 x0 + w1 x1 + w0 x2 + w2 x0]_k): the three dropped
// products (w1 x2, w2 x1, w2 x2) are <= 2^-24 |w x| each, the kept ones are exact in the matrix pipe, the sums are fp32 -- the
// result differs from the exact k-ascending fmaf chain (the metric's arithmetic, csrc/sa_pipe.hip) by fp32-roundoff-sized terms
// (tests/test_x6_gpu.py: <= 2e-6 of the layer's largest output), and NOT bit for bit.  The first layer's xyz part (K = 3..6) stays on
// v_mfma_f32_32x32x2_f32: exact, and cheaper than six padded bf16 products.
//
// Why it is its own kernel and not sa_bf16.hip with three operands: six MFMAs per k-step change the balance.
//  * A 32-position tile's activations are 6 bytes per element: 128 + 208 channels of one tile are 252 registers, so a wave (one per
//    SIMD, 512 registers) owns ONE tile, not four, and a weight fragment is used by six MFMAs of one tile instead of one MFMA of four.
//  * Weights therefore cannot be streamed per wave (3 KiB per 192 matrix-pipe cycles and wave = 64 B/clk per CU, the whole L1 rate):
//    the four waves of a workgroup share them through LDS.  SA2 (480 KB of split fragments): a ring of three slots filled by
//    LDS-DMA (global_load_lds_dwordx4, one row tile = one chunk, every wave issues a quarter), one s_barrier per chunk; SA1
//    (36-116 KB): the whole image resident, no barrier after start-up.
//  * ZERO-SWAP HAND-OVER as in sa_bf16.hip (the accumulator tile is the next layer's operand when the weights' k order is
//    permuted), the split is 12 VALU per register pair; the LAST LAYER IS FLIPPED (positions = rows, a lane owns a channel: the max
//    over the neighbours is 8 v_max3 per tile), bias and ReLU after the max.
//  * every epilogue is deferred by one tile and spread over the next tile's k-steps behind its MFMAs.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void sx_lds_void;
typedef const __attribute__((address_space(1))) void sx_glb_void;

constexpr int sx_cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int sx_pad4(int a) { return (a + 3) / 4 * 4; }
constexpr int sx_max(int a, int b) { return a > b ? a : b; }
constexpr int sx_pad128(int c) { return (c + 127) / 128 * 128; }

__device__ __forceinline__ unsigned sx_pack(float lo, float hi) {          // one v_cvt_pk_bf16_f32 (RNE)
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float sx_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float sx_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// (a, b) -> the packed pairs of their three parts
__device__ __forceinline__ void sx_split2(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
    p0 = sx_pack(a, b);
    const float ra = a - sx_lo(p0), rb = b - sx_hi(p0);
    p1 = sx_pack(ra, rb);
    p2 = sx_pack(ra - sx_lo(p1), rb - sx_hi(p1));
}

__device__ __forceinline__ f32x16 sx_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of one k-step as TWO accumulator chains issued alternately -- `lo` takes the three small products (2^-16 of the
// leading one), `hi` the three large ones; a tile's result is hi + lo.  Back-to-back MFMAs on ONE accumulator run at the issue rate only
// while nothing sits between them: every ds_read / VALU the scheduler puts into such a chain costs ~43 cycles (MI355X_MICROARCH.md),
// and a k-step carries three fragment reads -- one chain ran the matrix pipe at 0.62-0.68; between MFMAs on different accumulators a
// filler costs ~6.  FLIP: activations are the A operand (rows = positions)
template <bool FLIP>
__device__ __forceinline__ void sx_group(f32x16 &hi, f32x16 &lo, const u32x4 (&w)[3], const u32x4 &x0, const u32x4 &x1, const u32x4 &x2) {
    if constexpr (FLIP) {
        lo = sx_mfma(x0, w[2], lo); hi = sx_mfma(x0, w[1], hi); lo = sx_mfma(x2, w[0], lo);
        hi = sx_mfma(x1, w[0], hi); lo = sx_mfma(x1, w[1], lo); hi = sx_mfma(x0, w[0], hi);
    } else {
        lo = sx_mfma(w[2], x0, lo); hi = sx_mfma(w[1], x0, hi); lo = sx_mfma(w[0], x2, lo);
        hi = sx_mfma(w[0], x1, hi); lo = sx_mfma(w[1], x1, lo); hi = sx_mfma(w[0], x0, hi);
    }
}
__device__ __forceinline__ float sx_max3(float a, float b, float c) {
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- the body sees I as a constant expression (the
// scheduling-group builtins want integer constants)
template <int I, int N, typename F>
__device__ __forceinline__ void sx_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sx_static_for<I + 1, N>(f);
    }
}

template <int CF, int C1, int C2, int C3, bool PRE>
struct SxShape {
    static constexpr int CIN1 = CF + 3;
    static constexpr int KS1 = PRE ? 2 : sx_cdiv(CF + 3, 2);        // fp32 k-steps (two rows each) of the first layer's MFMA part
    static constexpr int NT1 = sx_cdiv(C1, 32), NT2 = sx_cdiv(C2, 32), NT3 = sx_cdiv(C3, 32);
    static constexpr int KST2 = sx_cdiv(C1, 16), KST3 = sx_cdiv(C2, 16);
    // a CHUNK = one row tile of layer 2 / 3: its k-steps' fragment triples (k-step major: fragment 3 kk + part), padded to a
    // multiple of four fragments so that every wave of a workgroup issues the same number of LDS-DMA pieces
    static constexpr int CH2 = sx_pad4(KST2 * 3), CH3 = sx_pad4(KST3 * 3);
    static constexpr int NCH = NT2 + NT3;
    static constexpr int WBYTES = (NT2 * CH2 + NT3 * CH3) * 1024;
    static constexpr int SLOTB = sx_max(CH2, CH3) * 1024;
    static constexpr int B1OFF = 0, B2OFF = NT1 * 32, B3OFF = B2OFF + NT2 * 32, NBIAS = B3OFF + NT3 * 32;   // floats behind the fragments
    static constexpr int IMG_BYTES = WBYTES + NBIAS * 4;
    static constexpr int chunk_off(int c) { return (c < NT2 ? c * CH2 : NT2 * CH2 + (c - NT2) * CH3) * 1024; }
    static constexpr int chunk_frags(int c) { return c < NT2 ? CH2 : CH3; }
};

// ---- image builder ----------------------------------------------------------------------------------------------------------
// One thread per bf16 element of the fragment part, then the biases.  wt2 / wt3: packed fp32 W'^T, row-major part (element
// [k * ldw + cout]).  Fragment (chunk, kk, part): lane l = row 32 t + (l & 31), k-slots 8 (l >> 5) .. + 7 of k-step kk, slot s
// holding channel 16 kk + perm(s) (sa_bf16.hip's hand-over order).
struct SxPackParams {
    int c1, c2, c3, ldw2, ldw3;
    const float *b1, *wt2, *b2, *wt3, *b3;
    unsigned char *img;
};

__device__ __forceinline__ int sx_perm(int s) { return (s & 3) | ((s & 4) << 1) | ((s & 8) >> 1); }

__global__ void pack_sa_x6_kernel(SxPackParams p) {
    const int nt1 = (p.c1 + 31) / 32, nt2 = (p.c2 + 31) / 32, nt3 = (p.c3 + 31) / 32;
    const int kst2 = (p.c1 + 15) / 16, kst3 = (p.c2 + 15) / 16;
    const int ch2 = (kst2 * 3 + 3) / 4 * 4, ch3 = (kst3 * 3 + 3) / 4 * 4;
    const int nfrag = nt2 * ch2 + nt3 * ch3;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long long)nfrag * 512) {
        const int f = (int)(e >> 9), lane = (int)(e >> 3) & 63, el = (int)e & 7;
        const int slot = 8 * (lane >> 5) + el;
        float v = 0.f;
        int part = 0;
        if (f < nt2 * ch2) {
            const int t = f / ch2, r = f % ch2, kk = r / 3;
            part = r % 3;
            const int row = 32 * t + (lane & 31), k = 16 * kk + sx_perm(slot);
            if (kk < kst2 && row < p.c2 && k < p.c1) v = p.wt2[(size_t)k * p.ldw2 + row];
        } else {
            const int g = f - nt2 * ch2;
            const int t = g / ch3, r = g % ch3, kk = r / 3;
            part = r % 3;
            const int row = 32 * t + (lane & 31), k = 16 * kk + sx_perm(slot);
            if (kk < kst3 && row < p.c3 && k < p.c2) v = p.wt3[(size_t)k * p.ldw3 + row];
        }
        const __bf16 h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const __bf16 h2 = (__bf16)(r1 - (float)h1);
        const __bf16 h = part == 0 ? h0 : (part == 1 ? h1 : h2);
        reinterpret_cast<unsigned short *>(p.img)[e] = __builtin_bit_cast(unsigned short, h);
    } else {
        const int i = (int)(e - (long long)nfrag * 512);
        const int nb = (nt1 + nt2 + nt3) * 32;
        if (i < nb) {
            float *bias = reinterpret_cast<float *>(p.img + (size_t)nfrag * 1024);
            float v = 0.f;
            if (i < nt1 * 32) v = (p.b1 != nullptr && i < p.c1) ? p.b1[i] : 0.f;
            else if (i < (nt1 + nt2) * 32) v = (i - nt1 * 32 < p.c2) ? p.b2[i - nt1 * 32] : 0.f;
            else v = (i - (nt1 + nt2) * 32 < p.c3) ? p.b3[i - (nt1 + nt2) * 32] : 0.f;
            bias[i] = v;
        }
    }
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------------
struct SxParams {
    int b, n, m, k;
    const float *feat;      // (B,CF,N) fp32 (small-input scales) or null
    const float *v1pm;      // PRE: (B,N,C1) fp32 POINT-major = b1 + W1[feature rows] feat (exact fp32: captra_pointwise_mlp_pm)
    const float *xyz_cn;    // (B,3,N)
    const float *new_xyz;   // (B,M,3)
    const int *idx;         // (B,M,K)
    const float *w1;        // packed fp32 first layer (row-major part: rows CF .. CF+2 = relative xyz; all CF+3 rows when !PRE)
    const unsigned char *img;
    float *out;             // (B,out_ctotal,M)
    int out_ctotal, co_off;
};

// s_waitcnt with only the vector-memory counter set (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#define SX_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)

// ReLU + three-way split of unit u (k-step half jj = u >> 2, register pair i = u & 3) of output tile t -> hout[part][2 t + jj][i]
template <int KSTN>
__device__ __forceinline__ void sx_split_unit(const f32x16 &acc, int t, int u, u32x4 (&hout)[3][KSTN]) {
    const int jj = u >> 2, i = u & 3;
    if (2 * t + jj >= KSTN) return;
    const float a = relu_bits(acc[8 * jj + 2 * i]), b = relu_bits(acc[8 * jj + 2 * i + 1]);
    unsigned p0, p1, p2;
    sx_split2(a, b, p0, p1, p2);
    // pin the unit HERE: left alone, LLVM sinks the whole split (pure arithmetic) down to its first use -- the next layer's first
    // MFMA -- five tiles' worth of it end up in one block that no MFMA covers and their accumulators stay live until then (the
    // 196-wide scale spills); the scheduling barriers around a group do not bind IR-level sinking, an asm statement that
    // "modifies" the three results does.  (With ONE accumulator chain per tile the pinned units cost more than they hid: 384 -> 453 us.)
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));
    hout[0][2 * t + jj][i] = p0;
    hout[1][2 * t + jj][i] = p1;
    hout[2][2 * t + jj][i] = p2;
}

// the group of the NEXT tile's k-steps behind whose MFMAs unit u (0..7) of a deferred split epilogue is issued: none in group 0
// (the tile's last MFMA is still executing), spread over groups 1 .. KST - 1
constexpr int sx_unit_group(int u, int kst) { return kst > 1 ? 1 + u * (kst - 1) / 8 : 0; }
// layer 2's LAST tile may be read out under layer 3's first tile when every unit precedes the k-step that consumes it
constexpr bool sx_defer_last_ok(int nt2, int kst3) {
    for (int u = 0; u < 8; ++u)
        if (2 * (nt2 - 1) + (u >> 2) < kst3 && sx_unit_group(u, kst3) > 2 * (nt2 - 1) + (u >> 2)) return false;
    return kst3 > 1;
}

// VALU instructions of the deferred read-outs issued with group (layer 3?, tile t, k-step kk): what the MFMAs of the group are interleaved with
constexpr int sx_group_valu(bool l3, int t, int kk, int nt2, int kst2, int kst3, bool defer_last) {
    int n = 0;
    if (!l3 && t == 0 && kk + 1 >= 2 && kk + 1 < kst2) n += 52;
    if (!l3 && t > 0)
        for (int u = 0; u < 8; ++u) n += sx_unit_group(u, kst2) == kk ? 13 : 0;
    if (l3 && t == 0 && defer_last)
        for (int u = 0; u < 8; ++u) n += sx_unit_group(u, kst3) == kk ? 13 : 0;
    if (l3 && t > 0 && kk == (kst3 > 1 ? 1 : 0)) n += 8;
    return n;
}

template <int CF, int C1, int C2, int C3, bool PRE, bool RING, int WAVES>
__device__ __forceinline__ void sx_body(const SxParams &p, unsigned char *smem) {
    using S = SxShape<CF, C1, C2, C3, PRE>;
    constexpr int NS = 3;                               // ring slots: chunk g lives in slot g % NS (NCH % NS == 0: static per chunk)
    static_assert(!RING || (S::NCH % NS == 0 && WAVES == 4), "ring: chunks per slice a multiple of the slots, four waves");
    static_assert(C1 % 32 == 0 && S::KS1 <= 3, "first-layer width");
    constexpr int WL = RING ? NS * S::SLOTB : S::WBYTES;
    float *bias_lds = reinterpret_cast<float *>(smem + WL);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;

    // LDS-DMA of chunk c (static) into its slot: this wave's quarter of the fragments
    // The pieces are issued as asm statements: hipcc (ROCm 7.2) orders every LDS read it can see behind ALL LDS-DMA in flight
    // (s_waitcnt vmcnt(0) in front of the first ds_read after an issue -- the bias reads of the next tile: the prefetch would wait for
    // itself), and it keeps a 64-bit per-lane address per piece of the global form and spills them.  Buffer form: one VGPR of lane
    // offsets, the piece's place in the image as a scalar offset; m0 = the piece's LDS byte address (wave-uniform; the hardware adds
    // lane x 16).  The compiler's own vmcnt waits stay safe: loads return in order, pieces it does not know of only make it wait longer.
    typedef int sx_i32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long img_addr = reinterpret_cast<unsigned long long>(p.img);
    const sx_i32x4 wsrc = {(int)(unsigned)img_addr, (int)(unsigned)(img_addr >> 32), S::WBYTES, 0x00020000};
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned voff16 = lane * 16;
    auto issue_chunk = [&](int c) {
        const unsigned dst = lds0 + (c % NS) * S::SLOTB + wave * 1024;
        const unsigned soff = S::chunk_off(c) + wave * 1024;
#pragma unroll
        for (int i = 0; i < S::chunk_frags(c) / 4; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst + i * 4096), "v"(voff16), "s"(wsrc), "s"(soff + i * 4096) : "memory");
    };
    // chunk c becomes readable (and chunk c + 1 goes on its way into the slot chunk c - 2 left: every wave is past that chunk's
    // last MFMA when it arrives here, so no LDS wait is needed in front of the barrier)
    auto acquire = [&](int c) {
        if constexpr (RING) {
#if !defined(CAPTRA_SX_ABL) || CAPTRA_SX_ABL < 2          // (timing ablations, results wrong: 1 = no LDS-DMA after start-up, 2 = no barrier either)
            SX_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
#endif
#if !defined(CAPTRA_SX_ABL) || CAPTRA_SX_ABL < 1
            issue_chunk((c + 1) % S::NCH);
#endif
        }
    };
    auto wbase = [&](int c) -> const unsigned char * {
        return smem + (RING ? (c % NS) * S::SLOTB : S::chunk_off(c)) + lane * 16;
    };

    // ---- start-up: biases (and the resident image) into LDS, first chunk on its way ------------------------------------------
    if constexpr (RING) issue_chunk(0);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.img + (RING ? S::WBYTES : 0));
        uint4 *dst = reinterpret_cast<uint4 *>(smem + (RING ? WL : 0));
        constexpr int N16 = ((RING ? 0 : S::WBYTES) + S::NBIAS * 4) / 16;
        for (int e = tid; e < N16; e += WAVES * 64) dst[e] = src[e];
    }
    // first layer's fp32 A operands: W1'^T[row 2 j + h][32 t + col]
    float at[S::KS1][S::NT1];
    {
        constexpr int LDW1 = sx_pad128(C1);
#pragma unroll
        for (int j = 0; j < S::KS1; ++j)
#pragma unroll
            for (int t = 0; t < S::NT1; ++t) {
                const int a = 2 * j + h;                                   // input row of this lane's half
                const int row = PRE ? CF + a : a;
                const bool ok = PRE ? a < 3 : a < CF + 3;
                at[j][t] = (ok && 32 * t + col < C1) ? p.w1[(size_t)row * LDW1 + 32 * t + col] : 0.f;
            }
    }
    if constexpr (RING) SX_WAIT_VM(0);                                      // chunk 0 landed (this wave's pieces) ...
    __syncthreads();                                                       // ... and everybody's; biases / the resident image visible
    if constexpr (RING) issue_chunk(1);

    const int ncentres = p.b * p.m;
    const int nslices = p.k / 32;
    const int njobs = (ncentres + WAVES - 1) / WAVES;
    float b3r[S::NT3];
#pragma unroll
    for (int t = 0; t < S::NT3; ++t) b3r[t] = bias_lds[S::B3OFF + 32 * t + col];

    // ---- a slice's per-lane inputs (this lane: position `col` of the slice, input rows of half h) ------------------------------
    auto load_id = [&](int c, int sl) -> int { return p.idx[(size_t)c * p.k + sl * 32 + col]; };
    // The gathered input rows arrive RAW (value, what to subtract from it) and become layer 1's operand at the START of the slice that
    // uses them: branch-free loads whose first use lies a whole layer 3 away.  (Written as `v = row[id] - centre` under `if (has_next)`,
    // load and subtraction shared a basic block and a scheduling region: s_waitcnt vmcnt(0) straight behind every gather, in the middle
    // of layers 2 / 3 -- three exposed round trips per slice.)
    auto load_bt = [&](int c, int id_, float (&raw_)[S::KS1], float (&sub_)[S::KS1]) {
        const int tb = c / p.m;
        const float *cp = p.new_xyz + (size_t)c * 3;
#pragma unroll
        for (int j = 0; j < S::KS1; ++j) {
            const int a = 2 * j + h;
            if constexpr (PRE) {
                // (the ring kernels keep the direct form: their chunk hand-over waits for every load in flight anyway, and the 196-wide
                // scale has no registers to spare for the raw pair)
                raw_[j] = a < 3 ? p.xyz_cn[((size_t)tb * 3 + a) * p.n + id_] - cp[a] : 0.f;
                sub_[j] = 0.f;
            } else {
                const int ax = a - CF < 0 ? 0 : (a - CF > 2 ? 2 : a - CF);
                const float *xrow = p.xyz_cn + ((size_t)tb * 3 + ax) * p.n;
                const float *row = xrow;
                if constexpr (CF > 0) row = a < CF ? p.feat + ((size_t)tb * CF + a) * p.n : xrow;
                raw_[j] = row[id_];
                sub_[j] = cp[ax];
            }
        }
    };
    auto finish_bt = [&](const float (&raw_)[S::KS1], const float (&sub_)[S::KS1], float (&bt_)[S::KS1]) {
#pragma unroll
        for (int j = 0; j < S::KS1; ++j) {
            const int a = 2 * j + h;
            if constexpr (PRE) bt_[j] = raw_[j];
            else bt_[j] = a < CF ? raw_[j] : (a < CF + 3 ? raw_[j] - sub_[j] : 0.f);
        }
    };
    constexpr int NG4 = PRE ? S::NT1 : 1;
    auto load_g4 = [&](int c, int id_, float4 (&g_)[NG4][4]) {
        if constexpr (PRE) {
            const int tb = c / p.m;
            const float4 *vp = reinterpret_cast<const float4 *>(p.v1pm + ((size_t)tb * p.n + id_) * C1 + 4 * h);
#pragma unroll
            for (int t = 0; t < S::NT1; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) g_[t][q] = vp[8 * t + 2 * q];       // rows 32 t + 8 q + 4 h + (0..3)
        }
    };

    int job = blockIdx.x;
    int c = job * WAVES + wave, sl = 0;
    int ce = c < ncentres ? c : ncentres - 1;           // (a spare wave recomputes the last centre and stores nothing)
    int id = 0;
    float bt[S::KS1], braw[S::KS1], bsub[S::KS1];
    float4 g4[NG4][4];
    if (job < njobs) {
        id = load_id(ce, 0);
        load_bt(ce, id, braw, bsub);
        load_g4(ce, id, g4);
    }
    float z[S::NT3];
    u32x4 wr[2][3];                                     // weight fragment triples, double-buffered: group gi in wr[gi & 1]
    constexpr int G2 = S::NT2 * S::KST2, G3 = S::NT3 * S::KST3, G = G2 + G3;
    constexpr bool DEFER_LAST = sx_defer_last_ok(S::NT2, S::KST3);
    auto wload = [&](int gi, u32x4 (&dst)[3]) {         // group gi (static) of a slice: (chunk, k-step)
        const int cch = gi < G2 ? gi / S::KST2 : S::NT2 + (gi - G2) / S::KST3;
        const int kk = gi < G2 ? gi % S::KST2 : (gi - G2) % S::KST3;
        const unsigned char *bp = wbase(cch) + kk * 3072;
#pragma unroll
        for (int s = 0; s < 3; ++s) dst[s] = *reinterpret_cast<const u32x4 *>(bp + s * 1024);
    };
    if (job < njobs) wload(0, wr[0]);

    while (job < njobs) {
        if (sl == 0) {
#pragma unroll
            for (int t = 0; t < S::NT3; ++t) z[t] = -__builtin_inff();
        }
        // ---- what comes after this slice (wave-uniform) ----
        const bool last_slice = sl + 1 == nslices;
        const int jobn = last_slice ? job + (int)gridDim.x : job;
        const int sn = last_slice ? 0 : sl + 1;
        const int cn_raw = jobn * WAVES + wave;
        const int cn = cn_raw < ncentres ? cn_raw : ncentres - 1;
        const bool has_next = jobn < njobs;
        int id_n = 0;

        u32x4 h1[3][S::KST2], h2[3][S::KST3];
        finish_bt(braw, bsub, bt);
        // ---- layer 1 (fp32 MFMA, exact): from the bias / the gathered v1 rows ------------------------------------------------------
        f32x16 acc1[S::NT1];
#pragma unroll
        for (int t = 0; t < S::NT1; ++t) {
            if constexpr (PRE) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc1[t][4 * q + 0] = g4[t][q].x; acc1[t][4 * q + 1] = g4[t][q].y;
                    acc1[t][4 * q + 2] = g4[t][q].z; acc1[t][4 * q + 3] = g4[t][q].w;
                }
            } else {
                const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + S::B1OFF + 32 * t + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = bp[2 * q];
                    acc1[t][4 * q + 0] = v.x; acc1[t][4 * q + 1] = v.y; acc1[t][4 * q + 2] = v.z; acc1[t][4 * q + 3] = v.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < S::KS1; ++j)
#pragma unroll
            for (int t = 0; t < S::NT1; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[j][t], bt[j], acc1[t], 0, 0, 0);
        // k-steps 0 and 1 of layer 2's operand now, the later ones behind layer 2's first MFMAs (unit u of tile t1 -> k-step 2 t1 + (u >> 2))
#pragma unroll
        for (int u = 0; u < 8; ++u) sx_split_unit<S::KST2>(acc1[0], 0, u, h1);
        __builtin_amdgcn_sched_barrier(0);

        // ---- layers 2 and 3: G groups of six MFMAs; tile tau accumulates in acc[tau & 1], tile tau - 1 is read out behind them ------
        f32x16 acc[2], accl[2];                         // (hi, lo) chains of tile tau in acc[tau & 1], accl[tau & 1]; merged into acc at read-out
        sx_static_for<0, G>([&](auto gi_c) __attribute__((always_inline)) {
            constexpr int gi = decltype(gi_c)::value;
            constexpr bool l3 = gi >= G2;
            constexpr int t = l3 ? (gi - G2) / S::KST3 : gi / S::KST2;
            constexpr int kk = l3 ? (gi - G2) % S::KST3 : gi % S::KST2;
            constexpr int kst = l3 ? S::KST3 : S::KST2;
            constexpr int tau = l3 ? S::NT2 + t : t;
            // the next tile's chunk, one group ahead of its first fragment read
            if (kk == kst - 1) acquire((tau + 1) % S::NCH);
            // fragments of the next group (the next slice's first group behind the last one: the weights repeat)
            if (gi + 1 < G) wload(gi + 1, wr[(gi + 1) & 1]);
            else wload(0, wr[(gi + 1) & 1]);
            // the next slice's inputs: ids under layer 2, the gathers under layer 3 (resident-image kernels: always asked for -- behind the
            // last slice they are the clamped centre's again and nobody reads them; no branch, nothing of a load's use in its own region)
            if (gi == 1 && (!PRE || has_next)) id_n = load_id(cn, sn);
            if (gi == G2 + 1 && (!PRE || has_next)) {
                // (the id becomes visible HERE: its sign extension for the 64-bit addresses otherwise sits straight behind the load, a wait
                // for the round trip in the middle of layer 2)
                if constexpr (!PRE) asm volatile("" : "+v"(id_n));
                load_bt(cn, id_n, braw, bsub);
                load_g4(cn, id_n, g4);
            }
            if (kk == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accl[tau & 1][r] = 0.f;
                if (!l3) {
                    const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + S::B2OFF + 32 * t + 4 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = bp[2 * q];
                        acc[tau & 1][4 * q + 0] = v.x; acc[tau & 1][4 * q + 1] = v.y; acc[tau & 1][4 * q + 2] = v.z; acc[tau & 1][4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tau & 1][r] = 0.f;
                }
            }
            // ---- deferred read-outs issued with this group ----
            if constexpr (tau > 0 && kk == (kst > 1 ? 1 : 0) && !(l3 && t == 0 && !DEFER_LAST))
                acc[(tau - 1) & 1] += accl[(tau - 1) & 1];                 // the finished tile: hi + lo
            if constexpr (!l3 && t == 0) {
                // layer 1's later tiles: the four units of k-step kk + 1
                constexpr int ks = kk + 1;
                if constexpr (ks >= 2 && ks < S::KST2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sx_split_unit<S::KST2>(acc1[ks >> 1], ks >> 1, 4 * (ks & 1) + i, h1);
                }
            }
            if constexpr (!l3 && t > 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (sx_unit_group(u, S::KST2) == kk) { sx_split_unit<S::KST3>(acc[(tau - 1) & 1], t - 1, u, h2); }
            }
            if constexpr (l3 && t == 0 && DEFER_LAST) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (sx_unit_group(u, S::KST3) == kk) { sx_split_unit<S::KST3>(acc[(tau - 1) & 1], S::NT2 - 1, u, h2); }
            }
            if constexpr (l3 && t > 0 && kk == (S::KST3 > 1 ? 1 : 0)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) z[t - 1] = sx_max3(z[t - 1], acc[(tau - 1) & 1][2 * i], acc[(tau - 1) & 1][2 * i + 1]);
            }
            // ---- the six products ----
            if (!l3) sx_group<false>(acc[tau & 1], accl[tau & 1], wr[gi & 1], h1[0][kk], h1[1][kk], h1[2][kk]);
            else sx_group<true>(acc[tau & 1], accl[tau & 1], wr[gi & 1], h2[0][kk], h2[1][kk], h2[2][kk]);
            // one MFMA, then a sixth of the group's VALU work: each of them issues while an MFMA executes
            {
                constexpr int per = (sx_group_valu(l3, t, kk, S::NT2, S::KST2, S::KST3, DEFER_LAST) + 5) / 6;
                if constexpr (per > 0) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // layer 2's last tile, when it cannot be read out under layer 3
            if constexpr (!l3 && gi == G2 - 1 && !DEFER_LAST) {
                acc[tau & 1] += accl[tau & 1];
#pragma unroll
                for (int u = 0; u < 8; ++u) sx_split_unit<S::KST3>(acc[tau & 1], S::NT2 - 1, u, h2);
            }
        });
        {
            constexpr int TL = S::NCH - 1;
            acc[TL & 1] += accl[TL & 1];
#pragma unroll
            for (int i = 0; i < 8; ++i) z[S::NT3 - 1] = sx_max3(z[S::NT3 - 1], acc[TL & 1][2 * i], acc[TL & 1][2 * i + 1]);
        }
        id = id_n;
        // ---- the centre's maxima: join the half-waves, bias, ReLU, store ------------------------------------------------------------
        if (last_slice) {
            const int tb = ce / p.m, centre = ce - tb * p.m;
#pragma unroll
            for (int t = 0; t < S::NT3; ++t) {
                const unsigned u = __float_as_uint(z[t]);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) + b3r[t];
                v = v > 0.f ? v : 0.f;
                const int ch = 32 * t + col;
                if (h == 0 && ch < C3 && c < ncentres) p.out[((size_t)tb * p.out_ctotal + p.co_off + ch) * p.m + centre] = v;
            }
        }
        job = jobn; sl = sn; c = cn_raw; ce = cn;
    }
    // nothing of the ring may land in LDS after the workgroup is gone
    if constexpr (RING) SX_WAIT_VM(0);
}

template <int CF, int C1, int C2, int C3, bool PRE, bool RING, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void sa_x6_kernel(SxParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sx_body<CF, C1, C2, C3, PRE, RING, WAVES>(p, smem);
}

template <int CF, int C1, int C2, int C3, bool PRE, bool RING, int WAVES>
int sx_launch(const SxParams &p, hipStream_t stream) {
    using S = SxShape<CF, C1, C2, C3, PRE>;
    const int lds = (RING ? 3 * S::SLOTB : S::WBYTES) + S::NBIAS * 4;
    auto kern = sa_x6_kernel<CF, C1, C2, C3, PRE, RING, WAVES>;
    static CaptraDeviceOnce once;
    if (lds > 48 * 1024 && once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    static std::atomic<int> cus_of[128];
    cus = cus_of[dev & 127].load(std::memory_order_relaxed);
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_of[dev & 127].store(cus, std::memory_order_relaxed);
    }
    const long long njobs = ((long long)p.b * p.m + WAVES - 1) / WAVES;
    const int per_cu = 1;                               // (the register budget is cut for WAVES / 4 waves per SIMD: one workgroup per CU)
    const unsigned grid = (unsigned)(njobs < (long long)cus * per_cu ? njobs : (long long)cus * per_cu);
    CAPTRA_LAUNCH("sa_scale_x6", kern, dim3(grid), dim3(WAVES * 64), lds, stream, p);
    return captra_last_error();
}

}  // namespace

extern "C" long long captra_sa_x6_image_bytes(int cfeat, int c1, int c2, int c3) {
    if (cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1) return -1;
    const long long nt1 = (c1 + 31) / 32, nt2 = (c2 + 31) / 32, nt3 = (c3 + 31) / 32;
    const long long ch2 = (((c1 + 15) / 16) * 3 + 3) / 4 * 4, ch3 = (((c2 + 15) / 16) * 3 + 3) / 4 * 4;
    return (nt2 * ch2 + nt3 * ch3) * 1024 + (nt1 + nt2 + nt3) * 32 * 4;
}

// wt2 / wt3, b1 / b2 / b3: the layers' PACKED fp32 buffers (include/captra_hip.h "PACKED WEIGHTS").  b1 may be NULL (pre-transformed
// first layer: the bias is inside v1).  The first layer's weights are not part of the image: the kernel reads its packed fp32 buffer.
extern "C" int captra_pack_sa_x6(int cfeat, int c1, int c2, int c3, const float *b1, const float *wt2, const float *b2,
                                 const float *wt3, const float *b3, unsigned char *img, captra_stream_t stream) {
    if (cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1 || wt2 == nullptr || wt3 == nullptr || b2 == nullptr || b3 == nullptr || img == nullptr) return -1;
    SxPackParams p;
    p.c1 = c1; p.c2 = c2; p.c3 = c3;
    p.ldw2 = (c2 + 127) / 128 * 128; p.ldw3 = (c3 + 127) / 128 * 128;
    p.b1 = b1; p.wt2 = wt2; p.b2 = b2; p.wt3 = wt3; p.b3 = b3; p.img = img;
    const long long total = captra_sa_x6_image_bytes(cfeat, c1, c2, c3);
    const long long nb = ((c1 + 31) / 32 + (c2 + 31) / 32 + (c3 + 31) / 32) * 32;
    const long long nfrag = (total - nb * 4) / 1024;
    const long long threads = nfrag * 512 + nb;
    CAPTRA_LAUNCH("pack_weights", pack_sa_x6_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return captra_last_error();
}

// One SA scale in the f32x6 arithmetic.  pre = 0: feat_or_v1 = feat (B,cfeat,N) fp32 or NULL (cfeat = 0), cfeat + 3 <= 6; pre = 1:
// feat_or_v1 = v1 (B,N,c1) fp32 POINT-major = b1 + W1[feature rows] feat.  w1: the first layer's packed fp32 buffer; img:
// captra_pack_sa_x6 of the same shape.  -2: shape not instantiated.
extern "C" int captra_sa_scale_x6(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                                  const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1, const unsigned char *img,
                                  float *out, int out_ctotal, int co_off, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1 || w1 == nullptr || img == nullptr) return -1;
    if (out_ctotal < co_off + c3 || co_off < 0) return -1;
    if (k % 32 != 0) return -2;
    if (b == 0 || m == 0) return 0;
    if ((long long)b * m * k >= (1ll << 31) || (long long)b * n * (pre ? c1 : 1) * 4 >= (1ll << 40) || (long long)b * m >= (1ll << 30)) return -2;
    if (pre && feat_or_v1 == nullptr) return -1;
    if (!pre && cfeat > 0 && feat_or_v1 == nullptr) return -1;
    SxParams p;
    p.b = b; p.n = n; p.m = m; p.k = k;
    p.feat = pre ? nullptr : feat_or_v1; p.v1pm = pre ? feat_or_v1 : nullptr;
    p.xyz_cn = xyz_cn; p.new_xyz = new_xyz; p.idx = idx; p.w1 = w1; p.img = img; p.out = out; p.out_ctotal = out_ctotal; p.co_off = co_off;
#define SX_CASE(CF_, C1_, C2_, C3_, PRE_, RING_, WAVES_)                                                         \
    if (cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_ && (pre != 0) == PRE_)                                \
        return sx_launch<CF_, C1_, C2_, C3_, PRE_, RING_, WAVES_>(p, (hipStream_t)stream);
    SX_CASE(320, 128, 128, 256, true, true, 4)
    SX_CASE(320, 128, 196, 256, true, true, 4)
    SX_CASE(0, 32, 32, 64, false, false, 8)
    SX_CASE(3, 32, 32, 64, false, false, 8)
    SX_CASE(0, 64, 64, 128, false, false, 8)
    SX_CASE(3, 64, 64, 128, false, false, 8)
    SX_CASE(0, 64, 96, 128, false, false, 8)
    SX_CASE(3, 64, 96, 128, false, false, 8)
#undef SX_CASE
    return -2;
}
