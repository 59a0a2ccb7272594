// bf16-native set-abstraction scale for gfx950 (BASELINE.json configs[2]: "bf16 shared-MLP on MFMA"): the body of the loop over
// radii of PointNetSetAbstractionMsg.forward (reference network/models/pointnet_utils.py:228-248: gather, centre subtraction,
// cat, 3 x (Conv2d 1x1 + BN + ReLU), max over K) as ONE launch on v_mfma_f32_32x32x16_bf16.
//
// Contract (include/captra_hip.h): every layer = act(b + sum_k bf16(w[k]) * bf16(x[k])), products exact, fp32 accumulation,
// weights rounded once at pack time (RNE), every layer's input rounded when it becomes an MFMA operand.
//
// What makes it a bf16 design rather than the fp32 kernel with the operand type swapped (round 2: 0.09 of the bf16 peak):
//  * ZERO-SWAP HAND-OVER.  The 32x32 accumulator tile (register r of lane l = row 8(r>>2)+(r&3)+4(l>>5), column l&31) is
//    the next layer's B operand as it stands if that layer's weights are stored with their K order PERMUTED: registers
//    8jj..8jj+7 of a lane become k-slots 8h..8h+7 of k-step 2t+jj, i.e. slot s of a 16-wide k-step holds channel
//    perm[s] = {0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15}.  The hand-over is 8 v_cvt_pk_bf16_f32 + 4 v_pk_max_i16 (ReLU on the
//    packed pairs: a negative bf16 is a negative int16) per 32x32 tile -- 12 instructions instead of 16 ReLU + 8
//    v_permlane32_swap + 8 cvt.
//  * THE LAST LAYER IS FLIPPED: activations are the A operand (rows = positions), weights the B operand (columns = output
//    channels), so a lane owns ONE output channel and the max over the K neighbours is a max over its 16 accumulator
//    registers (8 v_max3_f32) + one half-wave exchange per centre -- not a 45-instruction lane butterfly per tile.  The bias
//    is added after the max (x -> fl(x + b) is monotone, so max_k fl(s_k + b) = fl(max_k s_k + b)).
//  * EVERY WEIGHT FRAGMENT FEEDS TN = 2..4 POSITION TILES.  A bf16 MFMA consumes a 1 KiB A fragment every 32 cycles per SIMD;
//    one fragment per MFMA (round 2) is 128 B/clk per CU against an L1 that delivers 64.  The weights are a FRAGMENT
//    IMAGE (frag (t,kk): 64 lanes x 16 bytes, lane l = row 32t+(l&31), k-slots 8(l>>5)..+7) built once by
//    captra_pack_sa_bf16; the SA1 scales (6-39 KB) copy it to LDS once per workgroup and read it with conflict-free
//    ds_read_b128, the SA2 scales (98-164 KB) stream it from L2 with fully coalesced 1 KiB buffer loads, four tiles per load.
//  * A WAVE OWNS CG CONSECUTIVE CENTRES: running max in registers, results staged in a wave-private LDS strip and written
//    as 16-byte row segments (no 8x write amplification of 4-byte scattered stores).
//  * Layer 1 of the small-input scales carries its bias as two constant-one input channels (hi + lo bf16 split of b1:
//    2^-17 relative), so its accumulators start from the inline constant 0.
#include "common.h"

unsigned long long *captra_sa_prof_ptr();   // sa_fused.hip: the debug counters set by captra_sa_fused_set_prof

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32 (the vector fptrunc; two scalar casts compile to two conversions and a v_perm_b32)
__device__ __forceinline__ unsigned sb_pack(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
// ReLU on a packed bf16 pair
__device__ __forceinline__ unsigned sb_relu2(unsigned v) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
__device__ __forceinline__ f32x16 sb_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// max of three as ONE instruction, two forms.
//  ASM = false: v_maximum3_f32 from llvm.maximum (no canonicalising v_max_f32 x, x, x in front as fmaxf gets in IEEE mode).
//    The form of the kernels whose accumulators live in VGPRs: the first VALU read of an MFMA result needs software wait
//    states, which the compiler inserts for its own instructions and NOT inside asm statements (stale reads otherwise).
//  ASM = true: v_max3_f32 as an asm statement.  The form of the one-wave-per-SIMD kernels: their accumulators are AGPRs, the
//    operands reach the statement through compiler-issued v_accvgpr_read (hazards handled there), and the opaque statement
//    keeps the scheduler from interleaving the read-outs, which with llvm.maximum costs 700 bytes of scratch per lane.
template <bool ASM>
__device__ __forceinline__ float sb_max3(float a, float b, float c) {
    if constexpr (ASM) {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    } else {
        return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
    }
}

template <int CF, int C1, int C2, int C3, bool PRE>
struct SbShape {
    static constexpr int CIN1 = PRE ? 3 : CF + 3;                 // input rows of the MFMA part of layer 1
    static constexpr int NT1 = cdiv_c(C1, 32), NT2 = cdiv_c(C2, 32), NT3 = cdiv_c(C3, 32);
    static constexpr int KST2 = cdiv_c(C1, 16), KST3 = cdiv_c(C2, 16);
    static constexpr int F1 = 0, F2 = NT1, F3 = F2 + NT2 * KST2, NFRAG = F3 + NT3 * KST3;
    static constexpr int WBYTES = NFRAG * 1024;
    static constexpr int B2OFF = 0, B3OFF = NT2 * 32, NBIAS = (NT2 + NT3) * 32;   // floats behind the fragments
    static constexpr int IMG_BYTES = WBYTES + NBIAS * 4;
};

// The order in which a pass consumes the image's fragments with RG row tiles per accumulator group (k-step major inside a
// group): what the streamed-weight kernels prefetch along.
template <typename S, int RG>
struct SbUseOrder {
    int f[S::NFRAG];
    constexpr SbUseOrder() : f() {
        int u = 0;
        for (int t = 0; t < S::NT1; ++t) f[u++] = S::F1 + t;
        for (int tg = 0; tg < S::NT2; tg += RG)
            for (int kk = 0; kk < S::KST2; ++kk)
                for (int r = 0; r < RG && tg + r < S::NT2; ++r) f[u++] = S::F2 + (tg + r) * S::KST2 + kk;
        for (int tg = 0; tg < S::NT3; tg += RG)
            for (int kk = 0; kk < S::KST3; ++kk)
                for (int r = 0; r < RG && tg + r < S::NT3; ++r) f[u++] = S::F3 + (tg + r) * S::KST3 + kk;
    }
};

// ---- image builder --------------------------------------------------------------------------------------------------
// One thread per bf16 element of the fragment part, then the biases.  wt*: packed fp32 W'^T (row-major part: element
// [k * ldw + cout]).  Layer 1: rows [row0, row0 + cin1) of wt1 (PRE: the three xyz rows behind the feature rows) and, when
// fold_b1, bias b1 as rows cin1 (hi) and cin1 + 1 (lo).  Layers 2 / 3: K order permuted (header).
struct SbPackParams {
    int cin1, row0, fold_b1, c1, c2, c3, ldw1, ldw2, ldw3;
    const float *wt1, *b1, *wt2, *b2, *wt3, *b3;
    unsigned char *img;
};

__device__ __forceinline__ int sb_perm(int s) { return (s & 3) | ((s & 4) << 1) | ((s & 8) >> 1); }

__global__ void pack_sa_bf16_kernel(SbPackParams p) {
    const int nt1 = (p.c1 + 31) / 32, nt2 = (p.c2 + 31) / 32, nt3 = (p.c3 + 31) / 32;
    const int kst2 = (p.c1 + 15) / 16, kst3 = (p.c2 + 15) / 16;
    const int f2 = nt1, f3 = f2 + nt2 * kst2, nfrag = f3 + nt3 * kst3;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nfrag * 512) {
        const int f = e >> 9, lane = (e >> 3) & 63, el = e & 7;
        const int slot = 8 * (lane >> 5) + el;
        float v = 0.f;
        if (f < f2) {
            const int row = 32 * f + (lane & 31);
            if (row < p.c1) {
                if (slot < p.cin1) v = p.wt1[(size_t)(p.row0 + slot) * p.ldw1 + row];
                else if (p.fold_b1 && slot < p.cin1 + 2) {
                    const float b = p.b1[row];
                    const float hi = (float)(__bf16)b;
                    v = slot == p.cin1 ? hi : b - hi;
                }
            }
        } else if (f < f3) {
            const int t = (f - f2) / kst2, kk = (f - f2) % kst2;
            const int row = 32 * t + (lane & 31), k = 16 * kk + sb_perm(slot);
            if (row < p.c2 && k < p.c1) v = p.wt2[(size_t)k * p.ldw2 + row];
        } else {
            const int t = (f - f3) / kst3, kk = (f - f3) % kst3;
            const int row = 32 * t + (lane & 31), k = 16 * kk + sb_perm(slot);
            if (row < p.c3 && k < p.c2) v = p.wt3[(size_t)k * p.ldw3 + row];
        }
        const __bf16 h = (__bf16)v;
        reinterpret_cast<unsigned short *>(p.img)[e] = __builtin_bit_cast(unsigned short, h);
    } else {
        const int i = e - nfrag * 512;
        if (i < (nt2 + nt3) * 32) {
            float *bias = reinterpret_cast<float *>(p.img + (size_t)nfrag * 1024);
            bias[i] = i < nt2 * 32 ? (i < p.c2 ? p.b2[i] : 0.f) : (i - nt2 * 32 < p.c3 ? p.b3[i - nt2 * 32] : 0.f);
        }
    }
}

// ---- the kernel --------------------------------------------------------------------------------------------------------
struct SbParams {
    int n, m;
    const float *feat;      // (B,CF,N) fp32 (small-input scales) or null
    const float *v1pm;      // PRE: (B,N,C1) fp32 POINT-major = b1 + W1[feature rows] feat
    const float *xyz_cn;    // (B,3,N)
    const float *new_xyz;   // (B,M,3)
    const int *idx;         // (B,M,K)
    const unsigned char *img;
    float *out;             // (B,out_ctotal,M)
    int out_ctotal, co_off;
    int jobs_per_cloud, njobs;
    int m0, mhi;                // centres [m0, mhi) of every cloud (captra_set_centre_window; default 0, m)
    unsigned long long *prof;   // sa2_bf16_kernel: phase timers of a sample of waves (captra_sa_fused_set_prof), or null
};

// output tile (rows 32t.., this wave's 32 columns) -> next layer's B operands hout[2t], hout[2t+1]: ReLU + round + pack
template <int NOUT>
__device__ __forceinline__ void sb_mid_epilogue(const f32x16 &acc, int t, u32x4 (&hout)[NOUT]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        if (2 * t + jj >= NOUT) continue;
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = sb_relu2(sb_pack(acc[8 * jj + 2 * i], acc[8 * jj + 2 * i + 1]));
        hout[2 * t + jj] = v;
    }
}

template <bool ASM>
__device__ __forceinline__ float sb_reduce16(float z, const f32x16 &a) {
#pragma unroll
    for (int i = 0; i < 8; ++i) z = sb_max3<ASM>(z, a[2 * i], a[2 * i + 1]);
    return z;
}

// (WLDS kernels: at most 256 registers per lane asked for, which also keeps the accumulators in VGPRs -- with the whole
// 512-register file allowed the compiler puts them in AGPRs and every epilogue element costs a v_accvgpr_read first)
// OCC: waves per SIMD the register budget is cut for; RGS: row tiles per accumulator group (0: 1 for TN >= 4, else 2);
// PF (small-input scales): the NEXT pass's gather in flight under this pass's MFMAs -- neighbour ids two passes ahead, raw
// coordinates / features one pass ahead (a pass of the SA1 scales is 1-2 us of MFMA work behind two dependent global-memory
// latencies; same arithmetic, same bits)
// The kernel's body as a device function: `job0` = the workgroup's first job (its four waves take job0 .. job0 + 3).  The one-shot
// kernel below passes blockIdx.x * 4; the level-1 stream kernel (end of this file) calls it once per (window, scale, network)
// ticket with the window's four jobs.  Ends without a trailing barrier: a caller that re-uses the LDS synchronises first.
template <int CF, int C1, int C2, int C3, int K, bool PRE, int TN, int CG, bool WLDS, int RGS = 0, bool PF = false, int DBG = 0, int LRD = 1>
__device__ __forceinline__ void sb_body(const SbParams &p, unsigned char *smem, int job0) {
    using S = SbShape<CF, C1, C2, C3, PRE>;
    constexpr int TPC = K / 32;                        // 32-position tiles per centre
    constexpr int CPP = TN > TPC ? TN / TPC : 1;       // centres per pass
    constexpr int PPC = TPC > TN ? TPC / TN : 1;       // passes per centre
    constexpr int NPASS = CG * TPC / TN;               // passes per job
    constexpr int RG = RGS > 0 ? RGS : (TN >= 4 ? 1 : 2);   // row tiles per accumulator group
    constexpr bool STAGE_OUT = CG >= 4;                // results through a wave-private LDS strip, written as 16-byte segments
    static_assert((CG * TPC) % TN == 0 && (PRE || TN == 2) && K % 32 == 0, "shape");
    static_assert(PRE || S::CIN1 + 2 <= 8, "first layer: inputs + two bias rows fit the lower half-wave's eight k-slots");
    constexpr int WL = WLDS ? S::WBYTES : 0;
    float *bias_lds = reinterpret_cast<float *>(smem + WL);
    float *ost_all = bias_lds + S::NBIAS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    // ---- stage the image (WLDS) and the biases --------------------------------------------------------------
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.img + (WLDS ? 0 : S::WBYTES));
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        constexpr int N16 = (WL + S::NBIAS * 4) / 16;
        for (int e = tid; e < N16; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    const int job = job0 + wave;
    if (job >= p.njobs) return;                        // (no barrier below)
    const int b = job / p.jobs_per_cloud;
    const int centre0 = p.m0 + (job % p.jobs_per_cloud) * CG;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.img, 0, S::WBYTES, 0x00020000);
    // streamed weights: a ring of RD fragments, loaded RD - 1 uses ahead of the MFMAs that read them (a fragment feeds TN
    // MFMAs = TN x 32 cycles; an L2 hit takes 500+ cycles under this load)
    // (LDS-resident weights, LRD > 1: the same ring over ds_read_b128 -- left to itself the compiler issues a fragment's read right in
    // front of the two MFMAs that consume it, and every k-step waits out the LDS latency)
    constexpr int RD = WLDS ? LRD : 8;
    constexpr SbUseOrder<S, RG> ORDER{};
    u32x4 ring[RD];
    auto wload = [&](int f) -> u32x4 {
        if constexpr (WLDS) return *reinterpret_cast<const u32x4 *>(smem + ((DBG & 2) ? (f & 3) : f) * 1024 + lane * 16);
        else return __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, ((DBG & 2) ? (f & 7) : f) * 1024, 0);
    };
    int use = 0;                                       // compile-time after unrolling: position in ORDER
    auto wfrag = [&](int f) -> u32x4 {
        if constexpr (WLDS && RD == 1) return *reinterpret_cast<const u32x4 *>(smem + ((DBG & 2) ? (f & 3) : f) * 1024 + lane * 16);
        else {
            const u32x4 w = ring[use % RD];
            if (use + RD - 1 < S::NFRAG) ring[(use + RD - 1) % RD] = wload(ORDER.f[use + RD - 1]);
            ++use;
            return w;
        }
    };
    float b3r[S::NT3];
#pragma unroll
    for (int t = 0; t < S::NT3; ++t) b3r[t] = bias_lds[S::B3OFF + 32 * t + col];
    float z[S::NT3][CPP];
#pragma unroll
    for (int t = 0; t < S::NT3; ++t)
#pragma unroll
        for (int c = 0; c < CPP; ++c) z[t][c] = -__builtin_inff();
    float *ost = ost_all + wave * (CG * C3);
    const size_t cloud_idx = (size_t)b * p.m * K;
    const float *xb = p.xyz_cn + (size_t)b * 3 * p.n;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // small-input scales: the gather of pass ps (this lane's position: tile 2 ps + h, column col)
    auto gather_id = [&](int ps) -> int {
        const int tile = ps * 2 + h;
        int c = centre0 + tile / TPC;
        c = c < p.mhi ? c : p.mhi - 1;
        if constexpr (DBG & 1) return col + 32 * (tile & 3);
        return p.idx[cloud_idx + (size_t)c * K + (tile % TPC) * 32 + col];
    };
    // raw[0..CF) features, raw[CF..CF+3) coordinates, raw[CF+3..CF+6) the centre: loads only (the prefetching form keeps them
    // raw across the loop edge, so that nothing waits for them before the next pass begins)
    auto gather_raw = [&](int ps, int id, float (&raw)[CF + 6]) {
        const int tile = ps * 2 + h;
        int c = centre0 + tile / TPC;
        c = c < p.mhi ? c : p.mhi - 1;
        const float *cp = p.new_xyz + ((size_t)b * p.m + c) * 3;
        if constexpr (!PRE) {
#pragma unroll
            for (int k = 0; k < CF; ++k) raw[k] = p.feat[((size_t)b * CF + k) * p.n + id];
#pragma unroll
            for (int a = 0; a < 3; ++a) raw[CF + a] = xb[(size_t)a * p.n + id];
#pragma unroll
            for (int a = 0; a < 3; ++a) raw[CF + 3 + a] = cp[a];
        }
    };
    float pf_raw[CF + 6];
    int pf_id = 0;
    if constexpr (PF && !PRE) {
        const int id0 = gather_id(0);
        pf_id = gather_id(NPASS > 1 ? 1 : 0);
        gather_raw(0, id0, pf_raw);
    }

#pragma unroll 1
    for (int ps = 0; ps < NPASS; ++ps) {
        u32x4 x1[TN];
        int ids[TN];
        // ---- gather: neighbour ids, relative coordinates (+ features) as layer 1's B operands --------------
        if constexpr (!PRE) {
            // 64 lanes fetch 64 consecutive positions: lanes 0-31 tile 0, lanes 32-63 tile 1; one half-wave exchange per
            // register then leaves each tile's inputs in the lower half-wave (k-slots 0..7) over a zero upper half
            float raw[CF + 6], in[8];
            if constexpr (PF) {
                // unconditional (the last passes re-load their own position): no branch splits the pass's block
#pragma unroll
                for (int k = 0; k < CF + 6; ++k) raw[k] = pf_raw[k];
                gather_raw(ps + 1 < NPASS ? ps + 1 : NPASS - 1, pf_id, pf_raw);      // (pf_id: requested one pass ago)
                pf_id = gather_id(ps + 2 < NPASS ? ps + 2 : NPASS - 1);
            } else {
                gather_raw(ps, gather_id(ps), raw);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) in[k] = 0.f;
#pragma unroll
            for (int k = 0; k < CF; ++k) in[k] = raw[k];
#pragma unroll
            for (int a = 0; a < 3; ++a) in[CF + a] = raw[CF + a] - raw[CF + 3 + a];
            in[CF + 3] = 1.f;
            in[CF + 4] = 1.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (not swap(g, 0): hipcc 7.2 hoists the zero register out of the pass loop although the instruction
                // overwrites it, and the next pass swaps stale data in)
                const unsigned g = sb_pack(in[2 * i], in[2 * i + 1]);
                const auto sw = __builtin_amdgcn_permlane32_swap(g, g, false, false);     // (lo, lo), (hi, hi)
                x1[0][i] = h ? 0u : sw[0];
                x1[1][i] = h ? 0u : sw[1];
            }
            ids[0] = ids[1] = 0;
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int tile = ps * TN + j;
                int c = centre0 + tile / TPC;
                c = c < p.mhi ? c : p.mhi - 1;
                const int id = (DBG & 1) ? col + 32 * j : p.idx[cloud_idx + (size_t)c * K + (tile % TPC) * 32 + col];
                const float *cp = p.new_xyz + ((size_t)b * p.m + c) * 3;
                const float r0 = xb[id] - cp[0], r1 = xb[(size_t)p.n + id] - cp[1], r2 = xb[(size_t)2 * p.n + id] - cp[2];
                ids[j] = id;
                x1[j][0] = h ? 0u : sb_pack(r0, r1);
                x1[j][1] = h ? 0u : sb_pack(r2, 0.f);
                x1[j][2] = 0u;
                x1[j][3] = 0u;
            }
        }
        u32x4 h1[TN][S::KST2], h2[TN][S::KST3];
        if constexpr (RD > 1) {
            use = 0;
#pragma unroll
            for (int i = 0; i < RD - 1; ++i) ring[i] = wload(ORDER.f[i]);
        }
        // ---- layer 1 ---------------------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < S::NT1; ++t) {
            const u32x4 w = wfrag(S::F1 + t);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 acc;
                if constexpr (PRE) {
                    // accumulator start = gathered v1 (point-major: registers 4q..4q+3 = channels 32t + 8q + 4h + 0..3)
                    const float *vp = p.v1pm + ((size_t)b * p.n + ids[j]) * C1 + 32 * t + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(vp + 8 * q);
                        acc[4 * q + 0] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
                    }
                    acc = sb_mfma(w, x1[j], acc);
                } else {
                    acc = sb_mfma(w, x1[j], zero16);
                }
                sb_mid_epilogue<S::KST2>(acc, t, h1[j]);
            }
        }
        // ---- layer 2 ---------------------------------------------------------------------------------------
#pragma unroll
        for (int tg = 0; tg < S::NT2; tg += RG) {
            f32x16 bias[RG], acc[RG][TN];
#pragma unroll
            for (int r = 0; r < RG; ++r)
                if (tg + r < S::NT2) {
                    const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + S::B2OFF + 32 * (tg + r) + 4 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = bp[2 * q];
                        bias[r][4 * q + 0] = v.x; bias[r][4 * q + 1] = v.y; bias[r][4 * q + 2] = v.z; bias[r][4 * q + 3] = v.w;
                    }
                }
#pragma unroll
            for (int kk = 0; kk < S::KST2; ++kk)
#pragma unroll
                for (int r = 0; r < RG; ++r)
                    if (tg + r < S::NT2) {
                        const u32x4 w = wfrag(S::F2 + (tg + r) * S::KST2 + kk);
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[r][j] = sb_mfma(w, h1[j][kk], kk == 0 ? bias[r] : acc[r][j]);
                    }
#pragma unroll
            for (int r = 0; r < RG; ++r)
                if (tg + r < S::NT2)
#pragma unroll
                    for (int j = 0; j < TN; ++j) sb_mid_epilogue<S::KST3>(acc[r][j], tg + r, h2[j]);
        }
        // ---- layer 3, flipped: D[position][channel]; running max over the centre's positions ------------------
#pragma unroll
        for (int tg = 0; tg < S::NT3; tg += RG) {
            f32x16 acc[RG][TN];
#pragma unroll
            for (int kk = 0; kk < S::KST3; ++kk)
#pragma unroll
                for (int r = 0; r < RG; ++r)
                    if (tg + r < S::NT3) {
                        const u32x4 w = wfrag(S::F3 + (tg + r) * S::KST3 + kk);
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[r][j] = sb_mfma(h2[j][kk], w, kk == 0 ? zero16 : acc[r][j]);
                    }
#pragma unroll
            for (int r = 0; r < RG; ++r)
                if (tg + r < S::NT3)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int c = CPP > 1 ? j / (TN / CPP) : 0;
                        z[tg + r][c] = sb_reduce16<!WLDS>(z[tg + r][c], acc[r][j]);
                    }
        }
        // ---- a centre (or CPP centres) complete: join the half-waves, bias, ReLU, hand out ---------------------
        if (PPC == 1 || (ps % PPC) == PPC - 1) {
#pragma unroll
            for (int c = 0; c < CPP; ++c) {
                const int cl = PPC > 1 ? ps / PPC : ps * CPP + c;      // centre within the job
#pragma unroll
                for (int t = 0; t < S::NT3; ++t) {
                    const unsigned u = __float_as_uint(z[t][c]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) + b3r[t];
                    v = v > 0.f ? v : 0.f;
                    z[t][c] = -__builtin_inff();
                    const int ch = 32 * t + col;
                    if constexpr (STAGE_OUT) {
                        if (h == 0 && ch < C3) ost[cl * C3 + ch] = v;
                    } else {
                        if (h == 0 && ch < C3 && centre0 + cl < p.mhi && (!(DBG & 4) || v == 12345.f))
                            p.out[((size_t)b * p.out_ctotal + p.co_off + ch) * p.m + centre0 + cl] = v;
                    }
                }
            }
        }
    }
    if constexpr (STAGE_OUT && !(DBG & 4)) {
        // wave-private strip ost[centre][channel] -> out rows: lane l writes centres 4(l%(CG/4)).. of channel l/(CG/4) + ...
        constexpr int QPC = CG / 4;                    // 16-byte segments per channel row
        constexpr int CHS = 64 / QPC;                  // channels per sweep
        float *ob = p.out + ((size_t)b * p.out_ctotal + p.co_off) * p.m + centre0;
        const bool vec_ok = (p.m % 4) == 0 && (centre0 % 4) == 0 && centre0 + CG <= p.mhi;
#pragma unroll 1
        for (int c0 = 0; c0 < C3; c0 += CHS) {
            const int ch = c0 + lane / QPC, cq = (lane % QPC) * 4;
            if (ch < C3) {
                float4 v;
                v.x = ost[(cq + 0) * C3 + ch]; v.y = ost[(cq + 1) * C3 + ch];
                v.z = ost[(cq + 2) * C3 + ch]; v.w = ost[(cq + 3) * C3 + ch];
                float *dst = ob + (size_t)ch * p.m + cq;
                if (vec_ok) *reinterpret_cast<float4 *>(dst) = v;
                else {
                    if (centre0 + cq + 0 < p.mhi) dst[0] = v.x;
                    if (centre0 + cq + 1 < p.mhi) dst[1] = v.y;
                    if (centre0 + cq + 2 < p.mhi) dst[2] = v.z;
                    if (centre0 + cq + 3 < p.mhi) dst[3] = v.w;
                }
            }
        }
    }
}

template <int CF, int C1, int C2, int C3, int K, bool PRE, int TN, int CG, bool WLDS, int OCC = (WLDS ? 2 : 1), int RGS = 0, bool PF = false, int DBG = 0, int LRD = 1>
__global__ __launch_bounds__(256, OCC) void sa_bf16_kernel(SbParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sb_body<CF, C1, C2, C3, K, PRE, TN, CG, WLDS, RGS, PF, DBG, LRD>(p, smem, blockIdx.x * 4);
}

// ---- SA2 scales, second form: the epilogues UNDER the MFMAs ------------------------------------------------------------
// sa_bf16_kernel<320, ...> runs one wave per SIMD (its 4 x 128 positions of activations fill the register file), so nothing but
// the wave's own instruction stream can cover anything -- and that stream, as the compiler schedules it, is blocks of MFMAs
// (a weight fragment, four MFMAs, ...) followed by blocks of 110-130 accumulator reads / conversions / maxima: per pass 656
// MFMAs (21 k cycles) and ~2450 single-issue instructions nothing overlaps (~10 k cycles); ablations (captra_sa_bf16_set_variant,
// bits 4..: coalesced gather, eight fragments only, no stores) move the launch by 1-6 %: it is the stream, not the memory.
// Here every accumulator group exists TWICE: while row-tile group g + 1 accumulates, group g is read out -- one unit of four
// instructions (two accumulator reads, a conversion or a max3, a ReLU) behind each MFMA, the order pinned with
// sched_barrier(0) (an in-order wave hides at most ~5 single-issue instructions under a 32-cycle MFMA, MI355X_MICROARCH.md).
// Layer 2's last read-out runs under layer 3's first group, whose early k-steps do not need it.  The neighbour ids are
// requested before the biases are staged (one latency instead of two in front of the first MFMA).  Same MFMA sequence per
// position, same conversions: bit-identical to sa_bf16_kernel (tests/test_model_gpu.py).  Measured (v1 launch included): 123.2 ->
// 112.4 us (K = 128, 196 wide), 64.7 -> 58.2 us (K = 64) at 32 clouds; 67.2 -> 61.7 and 38.3 -> 35.3 at 16.  Less than the
// instruction count promised: the slots between MFMAs are not free on this part (MI355X_MICROARCH.md: a filler between two
// MFMAs costs 6 ... 20 cycles depending on where it lands), and a quarter of a wave's life is still the v1 gather in front of
// layer 1 (64 16-byte loads per lane, 32 cache lines each).
template <int C2, int K, int CG, int ILV = 1, bool PROF = false>
__global__ __launch_bounds__(256, 1) void sa2_bf16_kernel(SbParams p) {
    using S = SbShape<320, 128, C2, 256, true>;
    constexpr int C1 = 128, C3 = 256, TN = 4;
    constexpr int TPC = K / 32, CPP = TN / TPC;        // tiles per centre, centres per pass (K = 128: 1, K = 64: 2)
    static_assert(CG * TPC == TN && (K == 64 || K == 128), "one pass of four tiles per job");
    constexpr int RD = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    int job = blockIdx.x * 4 + wave;
    const bool live = job < p.njobs;
    job = live ? job : p.njobs - 1;                    // (a spare wave computes the last job again and stores nothing)
    // (PROF: phase timers of a sample of waves -- an instantiation of its own: the disabled timers alone cost the kernel a third)
    const bool sampled = PROF && blockIdx.x % 16 == 0 && lane == 0;
    const unsigned long long t_first = PROF ? __builtin_amdgcn_s_memtime() : 0ull, r_first = PROF ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long t_last = t_first;
#define SB2_TICK(slot)                                                                     \
    if constexpr (PROF) {                                                                  \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                     \
        if (sampled) atomicAdd(p.prof + (slot), t_now - t_last);                           \
        t_last = t_now;                                                                    \
    }
    const int b = job / p.jobs_per_cloud;
    const int centre0 = p.m0 + (job % p.jobs_per_cloud) * CG;
    const size_t cloud_idx = (size_t)b * p.m * K;
    const float *xb = p.xyz_cn + (size_t)b * 3 * p.n;
    // ---- neighbour ids and centres first, the biases behind them --------------------------------------------------------------
    int ids[TN];
    float ctr[TN][3];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int c = centre0 + j / TPC;
        c = c < p.mhi ? c : p.mhi - 1;
        ids[j] = p.idx[cloud_idx + (size_t)c * K + (j % TPC) * 32 + col];
        const float *cp = p.new_xyz + ((size_t)b * p.m + c) * 3;
        ctr[j][0] = cp[0]; ctr[j][1] = cp[1]; ctr[j][2] = cp[2];
    }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.img + S::WBYTES);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        constexpr int N16 = S::NBIAS * 4 / 16;
        for (int e = tid; e < N16; e += 256) dst[e] = src[e];
    }
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.img, 0, S::WBYTES, 0x00020000);
    constexpr SbUseOrder<S, 1> ORDER{};
    u32x4 ring[RD];
    auto wload = [&](int f) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, f * 1024, 0); };
    int use = 0;
    auto wfrag = [&]() -> u32x4 {                       // the fragments in ORDER (k-step major inside a row tile), RD - 1 uses ahead
        const u32x4 w = ring[use % RD];
        if (use + RD - 1 < S::NFRAG) ring[(use + RD - 1) % RD] = wload(ORDER.f[use + RD - 1]);
        ++use;
        return w;
    };
#pragma unroll
    for (int i = 0; i < RD - 1; ++i) ring[i] = wload(ORDER.f[i]);
    u32x4 x1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int id = ids[j];
        const float r0 = xb[id] - ctr[j][0], r1 = xb[(size_t)p.n + id] - ctr[j][1], r2 = xb[(size_t)2 * p.n + id] - ctr[j][2];
        x1[j][0] = h ? 0u : sb_pack(r0, r1);
        x1[j][1] = h ? 0u : sb_pack(r2, 0.f);
        x1[j][2] = 0u;
        x1[j][3] = 0u;
    }
    __syncthreads();
    SB2_TICK(0)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 h1[TN][S::KST2], h2[TN][S::KST3];
    // ---- layer 1: accumulators start from the gathered v1 rows, one k-step (the relative coordinates) ------------------------
#pragma unroll
    for (int t = 0; t < S::NT1; ++t) {
        const u32x4 w = wfrag();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x16 acc;
            const float *vp = p.v1pm + ((size_t)b * p.n + ids[j]) * C1 + 32 * t + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(vp + 8 * q);
                acc[4 * q + 0] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
            }
            acc = sb_mfma(w, x1[j], acc);
            sb_mid_epilogue<S::KST2>(acc, t, h1[j]);
        }
    }
    SB2_TICK(1)
    // ---- layers 2 and 3, two accumulator groups in flight -------------------------------------------------------------------------
    f32x16 acc[2][TN];
    float z[S::NT3][CPP];
#pragma unroll
    for (int t = 0; t < S::NT3; ++t)
#pragma unroll
        for (int c = 0; c < CPP; ++c) z[t][c] = -__builtin_inff();
    auto bias2 = [&](int tg) -> f32x16 {
        f32x16 bv;
        const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + S::B2OFF + 32 * tg + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = bp[2 * q];
            bv[4 * q + 0] = v.x; bv[4 * q + 1] = v.y; bv[4 * q + 2] = v.z; bv[4 * q + 3] = v.w;
        }
        return bv;
    };
    // read-out element i (0..3) of unit u = (tile u >> 1, half u & 1) of layer-2 group tg: two accumulator registers -> one packed pair
    auto epi2 = [&](int tg, int u, int i, const f32x16 (&a)[TN]) {
        const int j = u >> 1, jj = u & 1;
        if (2 * tg + jj < S::KST3) h2[j][2 * tg + jj][i] = sb_relu2(sb_pack(a[j][8 * jj + 2 * i], a[j][8 * jj + 2 * i + 1]));
    };
    // the same of layer-3 group tg: two registers of tile j into the running maximum of its centre
    auto epi3 = [&](int tg, int u, int i, const f32x16 (&a)[TN]) {
        const int j = u >> 1, hf = u & 1, c = CPP > 1 ? j / (TN / CPP) : 0;
        z[tg][c] = sb_max3<true>(z[tg][c], a[j][8 * hf + 2 * i], a[j][8 * hf + 2 * i + 1]);
    };
#pragma unroll
    for (int tg = 0; tg < S::NT2; ++tg) {
        const f32x16 bv = bias2(tg);
#pragma unroll
        for (int kk = 0; kk < S::KST2; ++kk) {
            const u32x4 w = wfrag();
#pragma unroll
            for (int j0 = 0; j0 < TN; j0 += ILV) {      // ILV MFMAs, then their ILV read-out units (ILV 1 / 2 / 4: 112.4 / 113.6 / 118.2 us at 32 clouds)
#pragma unroll
                for (int j = j0; j < j0 + ILV; ++j) acc[tg & 1][j] = sb_mfma(w, h1[j][kk], kk == 0 ? bv : acc[tg & 1][j]);
#pragma unroll
                for (int j = j0; j < j0 + ILV; ++j)
                    if (tg > 0 && kk < 8) epi2(tg - 1, kk, j, acc[(tg - 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    SB2_TICK(2)
    constexpr int P3 = S::NT2 & 1;                      // layer-3 group tg accumulates in acc[(tg + P3) & 1]: group 0 beside layer 2's last
    // layer 2's last group is read out under layer 3's first one -- when the k-steps it produces (2 NT2 - 2, 2 NT2 - 1) come after
    // the eight k-steps the read-out takes; otherwise (C2 = 128: k-steps 6, 7) before it
    constexpr bool DEFER_LAST = 2 * (S::NT2 - 1) >= 8;
    if constexpr (!DEFER_LAST) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) epi2(S::NT2 - 1, u, i, acc[(S::NT2 - 1) & 1]);
    }
#pragma unroll
    for (int tg = 0; tg < S::NT3; ++tg) {
#pragma unroll
        for (int kk = 0; kk < S::KST3; ++kk) {
            const u32x4 w = wfrag();
#pragma unroll
            for (int j0 = 0; j0 < TN; j0 += ILV) {
#pragma unroll
                for (int j = j0; j < j0 + ILV; ++j) acc[(tg + P3) & 1][j] = sb_mfma(h2[j][kk], w, kk == 0 ? zero16 : acc[(tg + P3) & 1][j]);
#pragma unroll
                for (int j = j0; j < j0 + ILV; ++j)
                    if (kk < 8) {
                        if (tg == 0) { if constexpr (DEFER_LAST) epi2(S::NT2 - 1, kk, j, acc[(S::NT2 - 1) & 1]); }
                        else epi3(tg - 1, kk, j, acc[(tg - 1 + P3) & 1]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) epi3(S::NT3 - 1, u, i, acc[(S::NT3 - 1 + P3) & 1]);
    SB2_TICK(3)
    // ---- the centres' maxima: join the half-waves, bias, ReLU, store -------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < CPP; ++c)
#pragma unroll
        for (int t = 0; t < S::NT3; ++t) {
            const unsigned u = __float_as_uint(z[t][c]);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) + bias_lds[S::B3OFF + 32 * t + col];
            v = v > 0.f ? v : 0.f;
            const int ch = 32 * t + col;
            if (live && h == 0 && ch < C3 && centre0 + c < p.m)
                p.out[((size_t)b * p.out_ctotal + p.co_off + ch) * p.m + centre0 + c] = v;
        }
    SB2_TICK(4)
    if (PROF && sampled) {
        atomicAdd(p.prof + 9, 1ull);
        atomicAdd(p.prof + 5, t_last - t_first);
        atomicAdd(p.prof + 6, __builtin_amdgcn_s_memrealtime() - r_first);
    }
#undef SB2_TICK
}

template <int C2, int K, int CG, int ILV = 1>
int sb2_launch(int b, SbParams p, hipStream_t stream) {
    if (CAPTRA_PROF_ON(p.prof) && ILV == 1) {
        using S = SbShape<320, 128, C2, 256, true>;
        p.jobs_per_cloud = (p.m + CG - 1) / CG;
        p.njobs = b * p.jobs_per_cloud;
        CAPTRA_LAUNCH("sa_scale_fused", (sa2_bf16_kernel<C2, K, CG, 1, true>), dim3((p.njobs + 3) / 4), dim3(256), S::NBIAS * 4, stream, p);
        return captra_last_error();
    }
    using S = SbShape<320, 128, C2, 256, true>;
    p.jobs_per_cloud = (p.m + CG - 1) / CG;
    p.njobs = b * p.jobs_per_cloud;
    CAPTRA_LAUNCH("sa_scale_fused", (sa2_bf16_kernel<C2, K, CG, ILV>), dim3((p.njobs + 3) / 4), dim3(256), S::NBIAS * 4, stream, p);
    return captra_last_error();
}

CAPTRA_KNOB int g_sb_variant = 0;      // A/B: bit 0 = small-input scales as before (no gather prefetch, no fragment ring); bit 3 = SA2 scales as before (sa_bf16_kernel); bits 4..: ablations (CAPTRA_ABLATIONS builds)

template <int CF, int C1, int C2, int C3, int K, bool PRE, int TN, int CG, bool WLDS, int OCC = (WLDS ? 2 : 1), int RGS = 0, bool PF = false, int DBG = 0, int LRD = 1>
int sb_launch(int b, SbParams p, hipStream_t stream) {
    using S = SbShape<CF, C1, C2, C3, PRE>;
    p.jobs_per_cloud = (p.mhi - p.m0 + CG - 1) / CG;
    p.njobs = b * p.jobs_per_cloud;
    if (p.njobs == 0) return 0;
    const int lds = (WLDS ? S::WBYTES : 0) + S::NBIAS * 4 + (CG >= 4 ? 4 * CG * C3 * 4 : 0);
    auto kern = sa_bf16_kernel<CF, C1, C2, C3, K, PRE, TN, CG, WLDS, OCC, RGS, PF, DBG, LRD>;
    static CaptraDeviceOnce once;
    if (lds > 48 * 1024 && once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    CAPTRA_LAUNCH("sa_scale_fused", kern, dim3((p.njobs + 3) / 4), dim3(256), lds, stream, p);
    return captra_last_error();
}

}  // namespace

extern "C" long long captra_sa_bf16_image_bytes(int cfeat, int c1, int c2, int c3) {
    if (c1 < 1 || c2 < 1 || c3 < 1) return -1;
    const long long nfrag = (c1 + 31) / 32 + (long long)((c2 + 31) / 32) * ((c1 + 15) / 16) + (long long)((c3 + 31) / 32) * ((c2 + 15) / 16);
    return nfrag * 1024 + ((c2 + 31) / 32 + (c3 + 31) / 32) * 32 * 4;
}

// wt1 / wt2 / wt3, b1 / b2 / b3: the layers' PACKED fp32 buffers (include/captra_hip.h "PACKED WEIGHTS").  pre != 0: the image's first
// layer holds the three xyz rows only (rows cfeat.. of wt1) and no bias (it is inside v1); else all cfeat + 3 rows and b1.
extern "C" int captra_pack_sa_bf16(int cfeat, int c1, int c2, int c3, int pre, const float *wt1, const float *b1, const float *wt2,
                                   const float *b2, const float *wt3, const float *b3, unsigned char *img, captra_stream_t stream) {
    if (cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1) return -1;
    if (!pre && cfeat + 3 + 2 > 8) return -2;
    SbPackParams p;
    p.cin1 = pre ? 3 : cfeat + 3; p.row0 = pre ? cfeat : 0; p.fold_b1 = pre ? 0 : 1;
    p.c1 = c1; p.c2 = c2; p.c3 = c3;
    p.ldw1 = (c1 + 127) / 128 * 128; p.ldw2 = (c2 + 127) / 128 * 128; p.ldw3 = (c3 + 127) / 128 * 128;
    p.wt1 = wt1; p.b1 = b1; p.wt2 = wt2; p.b2 = b2; p.wt3 = wt3; p.b3 = b3; p.img = img;
    const long long total = captra_sa_bf16_image_bytes(cfeat, c1, c2, c3);
    const long long nfrag = (total - ((c2 + 31) / 32 + (c3 + 31) / 32) * 128) / 1024;
    const long long threads = nfrag * 512 + ((c2 + 31) / 32 + (c3 + 31) / 32) * 32;
    CAPTRA_LAUNCH("pack_weights", pack_sa_bf16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return captra_last_error();
}

// One SA scale.  pre = 0: feat_or_v1 = feat (B,cfeat,N) fp32 (cfeat + 3 <= 6); pre = 1: feat_or_v1 = v1 (B,N,c1) fp32 POINT-major
// = b1 + W1[feature rows] feat.  img: captra_pack_sa_bf16 with the same (cfeat, c1, c2, c3, pre).
extern "C" int captra_sa_scale_bf16_ex(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                                    const float *xyz_cn, const float *new_xyz, const int *idx, const unsigned char *img, float *out,
                                    int out_ctotal, int co_off, const captra_launch_opts *opts, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1) return -1;
    if (out_ctotal < co_off + c3 || co_off < 0) return -1;
    if (b == 0 || m == 0) return 0;
    if ((long long)b * m * k >= (1ll << 31) || (long long)n * (pre ? c1 : 1) * 4 >= (1ll << 31)) return -2;
    SbParams p;
    p.n = n; p.m = m; p.feat = pre ? nullptr : feat_or_v1; p.v1pm = pre ? feat_or_v1 : nullptr;
    p.xyz_cn = xyz_cn; p.new_xyz = new_xyz; p.idx = idx; p.img = img; p.out = out; p.out_ctotal = out_ctotal; p.co_off = co_off;
    p.jobs_per_cloud = p.njobs = 0;
    p.prof = captra_sa_prof_ptr();
    {
        int wm0, wmc;
        const bool win = captra_centre_window(opts, m, &wm0, &wmc);
        if (win && pre) return -2;            // a centre window is the small-input scales' only (sa_bf16_kernel)
        p.m0 = wm0; p.mhi = wm0 + wmc;
    }
    hipStream_t s = (hipStream_t)stream;
#define SB_MATCH(CF_, C1_, C2_, C3_, K_, PRE_) (cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_ && k == K_ && (pre != 0) == PRE_)
    // PF_: the gather prefetch pays from 64 neighbours on (sa1s3 82 -> 79 us at 32 clouds, 54 -> 50 at 16; sa1s2 26 -> 23 at 16); the
    // K = 32 scale (four passes per wave) loses to its two extra loads
#define SB_CASE1(CF_, C1_, C2_, C3_, K_, PF_)                                                                \
    if (SB_MATCH(CF_, C1_, C2_, C3_, K_, false)) {                                                          \
        if (g_sb_variant & 1) return sb_launch<CF_, C1_, C2_, C3_, K_, false, 2, 8, true, 2, 0, false, 0, 1>(b, p, s);   \
        return sb_launch<CF_, C1_, C2_, C3_, K_, false, 2, 8, true, 2, 0, PF_, 0, 4>(b, p, s);              \
    }
#if CAPTRA_ABLATIONS
    if (SB_MATCH(0, 64, 96, 128, 128, false)) {
        switch (g_sb_variant >> 4) {
        case 1: return sb_launch<0, 64, 96, 128, 128, false, 2, 8, true, 2, 0, false, 1>(b, p, s);
        case 2: return sb_launch<0, 64, 96, 128, 128, false, 2, 8, true, 2, 0, false, 2>(b, p, s);
        case 3: return sb_launch<0, 64, 96, 128, 128, false, 2, 8, true, 2, 0, false, 3>(b, p, s);
        case 4: return sb_launch<0, 64, 96, 128, 128, false, 2, 8, true, 2, 0, false, 4>(b, p, s);
        case 7: return sb_launch<0, 64, 96, 128, 128, false, 2, 8, true, 2, 0, false, 7>(b, p, s);
        default: break;
        }
    }
#endif
    SB_CASE1(0, 32, 32, 64, 32, false) SB_CASE1(0, 64, 64, 128, 64, true) SB_CASE1(0, 64, 96, 128, 128, true)
    SB_CASE1(3, 32, 32, 64, 32, false) SB_CASE1(3, 64, 64, 128, 64, true) SB_CASE1(3, 64, 96, 128, 128, true)
    // (two tiles per wave on two waves per SIMD was measured for the SA2 scales: 63 -> 59 us for K = 64, and the 196-wide scale
    // does not fit 256 registers -- 500 bytes of scratch, 122 -> 155 us)
    if (SB_MATCH(320, 128, 128, 256, 64, true)) {
        if (!(g_sb_variant & 8)) return sb2_launch<128, 64, 2>(b, p, s);
        return sb_launch<320, 128, 128, 256, 64, true, 4, 2, false>(b, p, s);
    }
    if (SB_MATCH(320, 128, 196, 256, 128, true)) {
#if CAPTRA_ABLATIONS
        switch (g_sb_variant >> 4) {      // ablations (results wrong by construction): 1 coalesced gather, 2 eight fragments only, 4 no stores
        case 1: return sb_launch<320, 128, 196, 256, 128, true, 4, 1, false, 1, 0, false, 1>(b, p, s);
        case 2: return sb_launch<320, 128, 196, 256, 128, true, 4, 1, false, 1, 0, false, 2>(b, p, s);
        case 3: return sb_launch<320, 128, 196, 256, 128, true, 4, 1, false, 1, 0, false, 3>(b, p, s);
        case 4: return sb_launch<320, 128, 196, 256, 128, true, 4, 1, false, 1, 0, false, 4>(b, p, s);
        case 7: return sb_launch<320, 128, 196, 256, 128, true, 4, 1, false, 1, 0, false, 7>(b, p, s);
        default: break;
        }
#endif
        if (!(g_sb_variant & 8)) return sb2_launch<196, 128, 1>(b, p, s);
        return sb_launch<320, 128, 196, 256, 128, true, 4, 1, false>(b, p, s);
    }
#undef SB_CASE1
#undef SB_MATCH
    return -2;
}
extern "C" int captra_sa_scale_bf16(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                                    const float *xyz_cn, const float *new_xyz, const int *idx, const unsigned char *img, float *out,
                                    int out_ctotal, int co_off, captra_stream_t stream) {
    return captra_sa_scale_bf16_ex(b, n, m, k, cfeat, c1, c2, c3, pre, feat_or_v1, xyz_cn, new_xyz, idx, img, out, out_ctotal, co_off, nullptr, stream);
}

// Only the two forms the tests compare (bit 0: the small-input scales without gather prefetch / fragment ring, bit 3: the SA2 scales on
// sa_bf16_kernel) exist in the shipped library; the ablation instantiations (bits 4..: results wrong by construction) and the timing
// modes of the level-1 stream kernel are compiled only with -DCAPTRA_ABLATIONS=1 (CAPTRA_HIPCC_EXTRA): a stray CAPTRA_SA_BF16_VARIANT
// cannot select them.
extern "C" void captra_sa_bf16_set_variant(int v) { g_sb_variant = CAPTRA_ABLATIONS ? v : (v & 9); }

// ======================================================================================================================
// LEVEL-1 STREAM KERNEL: the 4096 -> 512 sampler, the ball query and the SA1 scales of the networks that share the cloud in ONE
// launch (reference pointnet_utils.py:214-249 PointNetSetAbstractionMsg.forward + sampling_gpu.cu:93-209 + ball_query_gpu.cu:9-45).
//
// The sampler is 511 dependent rounds on ONE workgroup per cloud: at 32 clouds, 0.24 ms with 224 of 256 CUs idle, and a fifth of
// the bf16 step (no schedule of separate launches hides it -- a window of centres is one round of work for kernels
// that lose B CUs to the sampler).  Here workgroups 0 .. B-1 run the sampler's loop (fps_round.h: the same rounds as
// fps_kernel_blocked<4, 16>) and PUBLISH every 8 picks as 8-byte {tag, pick} granules (one write-through store per lane: the data
// is the flag, MI355X_MICROARCH.md "R2"); every other workgroup -- and the sampler workgroups once they are done -- pulls TICKETS
// (window of 32 centres, scale, cloud) from one counter in window order, polls the window's granules, stages the cloud in LDS,
// runs the ball query of its scale for the window's centres (bq_scan.h: the same scan as ball_query_kernel) and then the scale's
// shared MLPs of every network on the lists it just wrote (sb_body: the same code as sa_bf16_kernel).  Nothing crosses workgroups
// but the granules: lists and centre coordinates are read back by the workgroup that wrote them.  Samplers have the lowest block
// ids (dispatched first) and wait for nobody, a ticket holder waits only for a sampler: no circular wait whatever the residency;
// every spin is bounded (ctl[1] != 0 afterwards = a consumer gave up).  Picks, lists and pooled features are bit-identical to
// captra_fps_gather + captra_ball_query_multi + 3 x captra_sa_scale_bf16 per network (tests/test_l1_stream_gpu.py).
// ======================================================================================================================
#include "bq_scan.h"
#include "fps_round.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long l1_gu64;
typedef __attribute__((address_space(1))) unsigned l1_gu32;

struct L1Net {
    const float *feat;                  // (B,CF,N) fp32, or null (CF = 0)
    const unsigned char *img[3];        // captra_pack_sa_bf16 images of the three scales
    float *out;                         // (B,out_ctotal,M)
    int out_ctotal, co_off[3];
};
struct L1Params {
    int b, n, m;
    const float *xyz_n3, *xyz_cn;
    const float *planes;                // (B,3,pad256(N)) the clouds in the ball query's LDS plane order (captra_bq_planes), or null
    int *fps_idx;
    float *new_n3, *new_cn;
    int m2;                             // level 2: the sampler workgroups go on to pick m2 of their m centres (0: not asked for)
    int *fps2_idx;                      // (B,m2) indices into the level-1 centres
    float *new2_n3, *new2_cn;           // (B,m2,3), (B,3,m2)
    int *idx[3];
    float r2[3];
    unsigned long long *gran;           // (B,M) granules {1, pick}; zeroed before the launch
    unsigned *ctl;                      // [0] ticket counter, [1] give-up flag; zeroed before the launch
    unsigned long long spin_limit;      // s_memrealtime ticks (100 MHz) a consumer waits for one window
    int prio;                           // s_setprio of the sampler's waves
    int nfine;                          // trailing centres handed out as fine windows of 8 (multiple of 32)
    int nwhole;                         // leading windows of 32 handed out whole (all three scales in one ticket)
    int pair;                           // scale tickets: 1 = two per window and cloud (scale 2 | scales 1 + 0 behind one scan), 0 = three
    int dbg;                            // timing experiments (results wrong): 1 = no ball-query scan, 2 = no MLPs, 4 = no cloud staging, 8 = no consumers, 16 = no sampling (picks of an earlier launch)
    L1Net net[2];
};

constexpr int L1_REGION_A = 57344;      // SA body (image + biases + strips <= 56192) | cloud planes (49152) | sampler (51456)
constexpr int L1_CTL = 1024;             // ticket words + the window's centres
constexpr int L1_LEVEL2 = 3 * 512 * 4 + 1024;   // the sampler's second level: its 512 centres' coordinates + 256 picks
constexpr int L1_LDS = L1_REGION_A + L1_CTL + L1_LEVEL2;

// one network's scale s on the window's four jobs (a wave = CG consecutive centres from c0 + CG * wave)
template <int CF, int C1, int C2, int C3, int K, bool PF, int CG>
__device__ __forceinline__ void l1_scale(const L1Params &p, const L1Net &net, int s, int b, int c0, int nc, unsigned char *smem) {
    SbParams q;
    q.n = p.n; q.m = p.m; q.feat = net.feat; q.v1pm = nullptr; q.xyz_cn = p.xyz_cn; q.new_xyz = p.new_n3; q.idx = p.idx[s];
    q.img = net.img[s]; q.out = net.out; q.out_ctotal = net.out_ctotal; q.co_off = net.co_off[s];
    q.jobs_per_cloud = 4; q.njobs = b * 4 + nc / CG; q.m0 = c0; q.mhi = p.m; q.prof = nullptr;      // (waves beyond the jobs leave after staging)
    sb_body<CF, C1, C2, C3, K, false, 2, CG, true, 0, PF, 0, 4>(q, smem, b * 4);
}

// the ball query of a ticket's centres [c0, c0 + nc) (nc / 4 per wave, four at a time against every group of the cloud: bq_scan.h) at
// the NR radii s0 .. s0 + NR - 1: one scan of the staged cloud whatever NR
template <int NR>
__device__ __forceinline__ void l1_ball_query(const L1Params &p, int s0, int b, int c0, int nc, unsigned char *smem, const float *ctr, int lane, int wave) {
    const float *xs = reinterpret_cast<const float *>(smem), *ys = xs + bq_pad(p.n), *zs = ys + bq_pad(p.n);
    constexpr int KS[3] = {32, 64, 128};
    float r2[NR];
    int ns[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { r2[r] = p.r2[s0 + r]; ns[r] = KS[0] << (s0 + r); }
    const int per_wave = (p.dbg & 1) ? 0 : nc / 4;
#pragma unroll 1
    for (int ci = 0; ci < per_wave; ci += 4) {
        float cx[4], cy[4], cz[4];
        int cnt[4][NR], first[4][NR];
        int *rows[NR];
        const int cl0 = per_wave * wave + ci;
#pragma unroll
        for (int r = 0; r < NR; ++r) rows[r] = p.idx[s0 + r] + ((size_t)b * p.m + c0 + cl0) * ns[r];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = ci + u < per_wave;
            const int cl = live ? cl0 + u : cl0;
            cx[u] = ctr[3 * cl + 0]; cy[u] = ctr[3 * cl + 1]; cz[u] = ctr[3 * cl + 2];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                cnt[u][r] = live ? 0 : ns[r];          // (a slot beyond the wave's centres: closed from the start, writes nothing)
                first[u][r] = 0;
            }
        }
        bq_scan_centres<4, NR>(xs, ys, zs, bq_pad(p.n) >> 8, 0, cx, cy, cz, r2, ns, rows, cnt, first, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (ci + u < per_wave) bq_pad_row(rows[r] + (size_t)u * ns[r], cnt[u][r], first[u][r], ns[r], lane);
    }
}

// the shared MLPs of scale s for the networks in `nets` (bit 0 / bit 1) on centres [c0, c0 + nc), CG centres per wave.  `next_ticket`
// non-null: this is the ticket's LAST scale -- the NEXT ticket is fetched in front of its last run and handed over behind it: the
// fetch's latency is off the path, and the ticket waits for this workgroup for one run only (fetched at the ticket's start it sat
// out the whole ticket while other workgroups idled at the end: consumers alone 350 -> 381 us at 32 clouds)
template <int CFA, int CFB, int C1, int C2, int C3, int K, bool PF, int CG>
__device__ __forceinline__ void l1_mlps(const L1Params &p, int s, int b, int c0, int nc, int nets, unsigned char *smem, int *next_ticket, int tid) {
    l1_gu32 *ctl = (l1_gu32 *)p.ctl;
    unsigned tn = 0u;
    const bool fetcher = tid == 0 && next_ticket != nullptr;
    const bool both = CFB >= 0 && nets == 3;
    if (((p.dbg & 2) || !both) && fetcher) tn = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(p.dbg & 2)) {
        if (nets & 1) l1_scale<CFA, C1, C2, C3, K, PF, CG>(p, p.net[0], s, b, c0, nc, smem);
        if constexpr (CFB >= 0) {
            if (both) {
                __syncthreads();
                if (fetcher) tn = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (nets & 2) l1_scale<CFB, C1, C2, C3, K, PF, CG>(p, p.net[1], s, b, c0, nc, smem);
        }
    }
    // (a workgroup gave up meanwhile: hand on the end-of-run sentinel, not a ticket fetched before the counter was lifted)
    if (fetcher) *next_ticket = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 0x7fffffff : (int)tn;
}

template <int CFA, int CFB>
__global__ __launch_bounds__(256, 2) void l1_stream_kernel(L1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_word = reinterpret_cast<int *>(smem + L1_REGION_A);          // [0], [1] tickets, [2] window ok
    float *ctr = reinterpret_cast<float *>(s_word + 4);                 // [32][3]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    l1_gu64 *gran = (l1_gu64 *)p.gran;
    l1_gu32 *ctl = (l1_gu32 *)p.ctl;
    // ctl[2] / [3] / [4]: earliest workgroup start (complemented), latest sampler end, latest workgroup end, in ticks of the 100 MHz counter
    // (read back by tools / tests: the sampler's span inside the launch, and the consumers' tail behind it)
    if (tid == 0) __hip_atomic_fetch_max(ctl + 2, ~(unsigned)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (max of the complement = the earliest)

    if ((int)blockIdx.x < p.b && (p.dbg & 16)) {
        // (timing experiment: the picks of an earlier launch, published at once -- the consumers' own throughput)
        for (int j = tid; j < p.m; j += 256)
            __hip_atomic_store(gran + (size_t)blockIdx.x * p.m + j, (1ull << 32) | (unsigned)p.fps_idx[(size_t)blockIdx.x * p.m + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if ((int)blockIdx.x < p.b) {
        // ---- sampler: cloud blockIdx.x, four waves x 16 points per lane, the rounds of fps_kernel_blocked<4, 16, true> --------------
        const int b = blockIdx.x, n = p.n, m = p.m;
        uint2 *slots = reinterpret_cast<uint2 *>(smem);                 // [2][16] ping-pong
        float *xs = reinterpret_cast<float *>(smem + 2 * 16 * sizeof(uint2));
        float *ys = xs + 4096, *zs = ys + 4096;
        int *picks = reinterpret_cast<int *>(zs + 4096);
        const float *xyz = p.xyz_n3 + (size_t)b * n * 3;
        const int base = tid * 16;
        fps_f32x2 px[8], py[8], pz[8];
        unsigned dmin[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = base + i;
            float x = 0.f, y = 0.f, z = 0.f;
            unsigned d0 = 0u;                  // slots beyond n: distance 0 forever, highest indices -> never preferred
            if (k < n) {
                x = xyz[(size_t)k * 3 + 0]; y = xyz[(size_t)k * 3 + 1]; z = xyz[(size_t)k * 3 + 2];
                d0 = __float_as_uint(1e10f);
                xs[k] = x; ys[k] = y; zs[k] = z;
            }
            px[i / 2][i & 1] = x; py[i / 2][i & 1] = y; pz[i / 2][i & 1] = z;
            dmin[i] = d0;
        }
        if (tid == 0) picks[0] = 0;
        __syncthreads();
        if (p.prio > 0) __builtin_amdgcn_s_setprio(3);
        int old = 0;
        // eight rounds, then their picks as one granule per lane (the wave's own LDS writes: in order): the round itself carries no
        // publishing code (inside the round it compiled to exec-masked LDS reads and a store on every round: 213 instructions for 171)
#pragma unroll 1
        for (int j8 = 0; j8 < m; j8 += 8) {
#pragma unroll 1
            for (int j = j8 > 0 ? j8 : 1; j < j8 + 8; ++j) {
                const float ox = xs[old], oy = ys[old], oz = zs[old];
                unsigned best, wmax, widx;
                int li;
                fps_lane_round16(px, py, pz, dmin, ox, oy, oz, best, li);
                fps_wave_winner(best, li, base, wmax, widx);
                uint2 *slot = slots + (j & 1) * 16;
                if (lane == 0) slot[wave] = make_uint2(wmax, widx);
                __syncthreads();
                old = fps_winner_of_four(slot);
                if (wave == 0) picks[j] = old;
            }
            if (wave == 0 && lane < 8)
                __hip_atomic_store(gran + (size_t)b * m + j8 + lane, (1ull << 32) | (unsigned)picks[j8 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.prio > 0) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        for (int j = tid; j < m; j += 256) p.fps_idx[(size_t)b * m + j] = picks[j];
        if (tid == 0) __hip_atomic_fetch_max(ctl + 3, (unsigned)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.m2 > 0) {
            // ---- level 2 (PointNet2Msg.sa2's sampling: m2 of the m centres just picked), while the consumers work off their backlog:
            // the rounds of fps_kernel_blocked<1, 8, true> on wave 0 (lane l holds centres 8l .. 8l + 7), the other waves wait ----
            float *c2x = reinterpret_cast<float *>(smem + L1_REGION_A + L1_CTL), *c2y = c2x + 512, *c2z = c2y + 512;
            int *picks2 = reinterpret_cast<int *>(c2z + 512);
            for (int j = tid; j < 512; j += 256) {
                const int id = j < m ? picks[j] : 0;
                c2x[j] = j < m ? xs[id] : 0.f; c2y[j] = j < m ? ys[id] : 0.f; c2z[j] = j < m ? zs[id] : 0.f;
            }
            __syncthreads();
            if (wave == 0) {
                fps_f32x2 qx[4], qy[4], qz[4];
                unsigned d2min[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = lane * 8 + i;
                    qx[i / 2][i & 1] = c2x[k]; qy[i / 2][i & 1] = c2y[k]; qz[i / 2][i & 1] = c2z[k];
                    d2min[i] = k < m ? __float_as_uint(1e10f) : 0u;
                }
                if (lane == 0) picks2[0] = 0;
                int old2 = 0;
#pragma unroll 1
                for (int j = 1; j < p.m2; ++j) {
                    unsigned best, wmax, widx;
                    int li;
                    fps_lane_round<8>(qx, qy, qz, d2min, c2x[old2], c2y[old2], c2z[old2], best, li);
                    fps_wave_winner(best, li, lane * 8, wmax, widx);
                    old2 = (int)widx;
                    picks2[j] = old2;
                }
            }
            __syncthreads();
            for (int j = tid; j < p.m2; j += 256) {
                const int id = picks2[j];
                p.fps2_idx[(size_t)b * p.m2 + j] = id;
                float *d3 = p.new2_n3 + ((size_t)b * p.m2 + j) * 3;
                d3[0] = c2x[id]; d3[1] = c2y[id]; d3[2] = c2z[id];
                float *dc = p.new2_cn + (size_t)b * 3 * p.m2 + j;
                dc[0] = c2x[id]; dc[p.m2] = c2y[id]; dc[2 * (size_t)p.m2] = c2z[id];
            }
        }
        __syncthreads();
    }

    // ---- consumer: tickets in window order -------------------------------------------------------------------------------------
    // Three kinds, so that the work is in few long tickets while the sampler has a long way to go and in many short ones at the end:
    //   WHOLE windows (the first p.nwhole windows of 32 centres): ticket = (window, cloud) -- ONE scan of the cloud for the three
    //     radii (the small ones never fill and walk all of it anyway), then the six MLP runs; one poll / staging per 32 x 3 lists;
    //   SCALE tickets (window, scale, cloud) for the windows behind them: a scan per scale, that scale's MLPs of every network;
    //   FINE tickets (window of 8, scale, network, cloud) for the last p.nfine centres: what is left when the sampler ends -- the
    //     backlog and the last window -- is many short tickets for the whole chip (a scale ticket at K = 128 is 110 us of one
    //     workgroup on an idle chip: 8 centres per wave through the scan, then 16 MLP passes per wave and network).
    constexpr int NNET = CFB >= 0 ? 2 : 1;
    const int wc = (p.m - p.nfine) / 32;               // windows of 32
    const int wa = p.nwhole < wc ? p.nwhole : wc;
    const int tpw = p.pair ? 2 : 3;                    // scale tickets per window and cloud: {2, 1, 0} or {2, 1 + 0}
    const int n_whole = wa * p.b, n_scale = n_whole + (wc - wa) * tpw * p.b;
    const int total = (p.dbg & 8) ? 0 : n_scale + (p.nfine / 8) * 3 * NNET * p.b;
    // s_word[0 / 1]: this ticket / the next one (l1_mlps fetches it); s_word[2]: window ok
    if (tid == 0) s_word[0] = total > 0 ? (int)__hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    for (int it = 0;; ++it) {
        __syncthreads();                               // (every wave is out of the previous ticket's LDS; the ticket word is there)
        const int t = s_word[it & 1];
        // (unsigned: once a workgroup gave up the counter sits at or above 0x7fffffff, which read as a signed ticket is negative)
        if ((unsigned)t >= (unsigned)total) break;
        int c0, nc, smask, b, nets;                    // first centre, centres, scales (bit s; the widest first), cloud, networks
        if (t < n_whole) {
            c0 = 32 * (t / p.b); nc = 32; smask = 7; b = t % p.b; nets = NNET == 2 ? 3 : 1;
        } else if (t < n_scale) {
            const int ts = t - n_whole;
            const int w = ts / (tpw * p.b), r = ts % (tpw * p.b), q = r / p.b;
            c0 = 32 * (wa + w); nc = 32; b = r % p.b; nets = NNET == 2 ? 3 : 1;
            smask = p.pair ? (q == 0 ? 4 : 3) : 1 << (2 - q);      // (pair: the two narrow scales share ONE scan of the cloud -- both walk all of it)
        } else {
            const int tf = t - n_scale, per = 3 * NNET * p.b;
            const int w = tf / per, r = tf % per;
            c0 = 32 * wc + 8 * w; nc = 8; smask = 1 << (2 - r / (NNET * p.b));
            const int r2 = r % (NNET * p.b);
            nets = 1 << (r2 / p.b); b = r2 % p.b;
        }
        if (wave == 0) {
            // wave 0: this window's granules -- one relaxed device-scope load per lane until every tag is there --, then the centres:
            // coordinates into LDS (ball query) and into new_xyz in both layouts
            // (the SA body and everything behind this launch read them there; the tickets of a window write the same values)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long g = 1ull << 32;
            bool ok;
            for (;;) {
                if (lane < nc) g = __hip_atomic_load(gran + (size_t)b * p.m + c0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = __all((g >> 32) == 1ull);
                if (ok || __builtin_amdgcn_s_memrealtime() - t0 > p.spin_limit) break;
                __builtin_amdgcn_s_sleep(16);
            }
            if (lane == 0) s_word[2] = ok ? 1 : 0;
            if (ok && lane < nc) {
                const int id = (int)(unsigned)g, c = c0 + lane;
                const float *q = p.xyz_n3 + ((size_t)b * p.n + id) * 3;
                const float x = q[0], y = q[1], z = q[2];
                ctr[3 * lane + 0] = x; ctr[3 * lane + 1] = y; ctr[3 * lane + 2] = z;
                float *d3 = p.new_n3 + ((size_t)b * p.m + c) * 3;
                d3[0] = x; d3[1] = y; d3[2] = z;
                float *dc = p.new_cn + (size_t)b * 3 * p.m + c;
                dc[0] = x; dc[p.m] = y; dc[2 * (size_t)p.m] = z;
            }
        } else if (!(p.dbg & 4)) {
            // waves 1-3: the cloud into the LDS planes (from the caller's plane image when there is one: straight 16-byte copies)
            float *xs = reinterpret_cast<float *>(smem), *ys = xs + bq_pad(p.n), *zs = ys + bq_pad(p.n);
            if (p.planes != nullptr) {
                const float4 *src = reinterpret_cast<const float4 *>(p.planes + (size_t)b * 3 * bq_pad(p.n));
                float4 *dst = reinterpret_cast<float4 *>(smem);
                const int n16 = 3 * bq_pad(p.n) / 4;
                for (int e = tid - 64; e < n16; e += 192) dst[e] = src[e];
            } else {
                bq_stage_tile(p.xyz_n3 + (size_t)b * p.n * 3, 0, p.n, xs, ys, zs, tid - 64, 192);
            }
        }
        __syncthreads();
        if (!s_word[2]) {
            // gave up: flag it and lift the ticket counter to a sentinel past every ticket, so that every other workgroup leaves at its
            // next fetch.  A maximum, not an addition: any number of workgroups giving up on the same stalled sampler leave the
            // counter at >= 0x7fffffff (an added 1 << 30 per workgroup wrapped to small tickets after four of them)
            if (tid == 0) {
                __hip_atomic_fetch_or(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_max(ctl, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
        if (smask == 7) l1_ball_query<3>(p, 0, b, c0, nc, smem, ctr, lane, wave);
        else if (smask == 3) l1_ball_query<2>(p, 0, b, c0, nc, smem, ctr, lane, wave);
        else l1_ball_query<1>(p, smask == 4 ? 2 : (smask == 2 ? 1 : 0), b, c0, nc, smem, ctr, lane, wave);
        int *next = s_word + ((it & 1) ^ 1);
#pragma unroll 1
        for (int s = 2; s >= 0; --s) {
            if (!((smask >> s) & 1)) continue;
            __syncthreads();                           // the lists are written (vmcnt drained), the planes / the last run's weights are free
            int *nx = (smask & ((1 << s) - 1)) == 0 ? next : nullptr;      // the ticket's last scale
            if (nc == 32) {
                if (s == 2) l1_mlps<CFA, CFB, 64, 96, 128, 128, true, 8>(p, s, b, c0, nc, nets, smem, nx, tid);
                else if (s == 1) l1_mlps<CFA, CFB, 64, 64, 128, 64, true, 8>(p, s, b, c0, nc, nets, smem, nx, tid);
                else l1_mlps<CFA, CFB, 32, 32, 64, 32, false, 8>(p, s, b, c0, nc, nets, smem, nx, tid);
            } else {
                if (s == 2) l1_mlps<CFA, CFB, 64, 96, 128, 128, true, 4>(p, s, b, c0, nc, nets, smem, nx, tid);
                else if (s == 1) l1_mlps<CFA, CFB, 64, 64, 128, 64, true, 4>(p, s, b, c0, nc, nets, smem, nx, tid);
                else l1_mlps<CFA, CFB, 32, 32, 64, 32, false, 4>(p, s, b, c0, nc, nets, smem, nx, tid);
            }
        }
    }
    if (tid == 0) __hip_atomic_fetch_max(ctl + 4, (unsigned)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the clouds in the ball query's LDS plane order (bq_scan.h), once per cloud instead of once per ticket
__global__ __launch_bounds__(256) void bq_planes_kernel(int n, const float *__restrict__ xyz_n3, float *__restrict__ planes) {
    const int b = blockIdx.y, npad = bq_pad(n);
    const int pt = blockIdx.x * 256 + threadIdx.x;
    if (pt >= npad) return;
    const bool inb = pt < n;
    const float *q = xyz_n3 + ((size_t)b * n + (inb ? pt : 0)) * 3;
    const float inf = __builtin_inff();
    const int chunk = pt >> 6;
    const int a = (((chunk >> 2) << 6) + (pt & 63)) * 4 + (chunk & 3);
    float *d = planes + (size_t)b * 3 * npad;
    d[a] = inb ? q[0] : inf;
    d[npad + a] = inb ? q[1] : inf;
    d[2 * npad + a] = inb ? q[2] : inf;
}

template <int CFA, int CFB>
int l1_launch(const L1Params &p, int grid, hipStream_t stream) {
    auto kern = l1_stream_kernel<CFA, CFB>;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    CAPTRA_LAUNCH("l1_stream", kern, dim3(grid), dim3(256), L1_LDS, stream, p);
    return captra_last_error();
}

CAPTRA_KNOB int g_l1_grid = 0;          // experiment knob: workgroups of the stream kernel (0 = one per CU up to 16 clouds, two beyond)
CAPTRA_KNOB int g_l1_prio = 1;
CAPTRA_KNOB int g_l1_fine = 32;
CAPTRA_KNOB int g_l1_whole = 0;
CAPTRA_KNOB int g_l1_pair = 1;
CAPTRA_KNOB int g_l1_dbg = 0;

}  // namespace

extern "C" void captra_sa1_stream_set_grid(int grid, int prio) { g_l1_grid = grid; g_l1_prio = prio; }
extern "C" void captra_sa1_stream_set_fine(int centres) { g_l1_fine = centres & 0xFFFF; g_l1_dbg = CAPTRA_ABLATIONS ? centres >> 16 : 0; }   // (bits 16..: timing ablations, results wrong: CAPTRA_ABLATIONS builds only)
extern "C" void captra_sa1_stream_set_whole(int windows) { g_l1_whole = windows & 0xFF; g_l1_pair = (windows >> 8) & 1 ? 0 : 1; }   // (bit 8: three scale tickets per window)

// planes (B,3,pad256(N)) <- xyz_n3 (B,N,3): element ((chunk / 4) * 64 + lane) * 4 + chunk % 4 of plane a = coordinate a of point
// 64 chunk + lane; slots beyond N hold +inf
extern "C" int captra_bq_planes(int b, int n, const float *xyz_n3, float *planes, captra_stream_t stream) {
    if (b < 0 || n < 1) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("bq_planes", bq_planes_kernel, dim3((bq_pad(n) + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, n, xyz_n3, planes);
    return captra_last_error();
}

extern "C" long long captra_sa1_stream_scratch_bytes(int b, int m) {
    if (b < 0 || m < 0) return -1;
    return (long long)b * m * 8 + 64;
}

// Level 1 of PointNet2Msg for the networks that share a cloud, in one launch: sampling (fps_idx, new_xyz in both layouts), the three
// ball queries (idx3[s] (B,M,K_s), K = 32 / 64 / 128) and the pooled features out_a / out_b (B,320,M) of the CAPTRA SA1 shapes
// [CF+3 -> 32 -> 32 -> 64], [-> 64 -> 64 -> 128], [-> 64 -> 96 -> 128] for input features feat_a (B,cfa,N) / feat_b (B,cfb,N), cf in {0, 3}
// (feat null for 0); cfb < 0: one network.  img_*: the scales' captra_pack_sa_bf16 images (pre = 0).  scratch:
// captra_sa1_stream_scratch_bytes(b, m) bytes, zeroed here on `stream`; after completion ((unsigned *)(scratch + b*m*8))[1] != 0
// means a consumer gave up waiting for the sampler (outputs incomplete).  -2: shape outside the kernel (n <= 4096, m <= 512, m % 32 == 0).
extern "C" int captra_sa1_stream_bf16(int b, int n, int m, const float *xyz_n3, const float *xyz_cn, const float *planes, const float *radius3, int *fps_idx,
                                      float *new_n3, float *new_cn, int *const *idx3, int cfa, const float *feat_a,
                                      const unsigned char *const *img_a3, float *out_a, int cfb, const float *feat_b,
                                      const unsigned char *const *img_b3, float *out_b, int m2, int *fps2_idx, float *new2_n3, float *new2_cn,
                                      void *scratch, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 1 || m2 < 0) return -1;
    if (n > 4096 || m > 512 || m % 32 || m > n || m2 > 256 || m2 > m) return -2;
    if (!((cfa == 0 || cfa == 3) && (cfb < 0 || cfb == 0 || cfb == 3))) return -2;
    if (b == 0) return 0;
    if (b > 256) return -2;
    hipStream_t st = (hipStream_t)stream;
    const long long sbytes = captra_sa1_stream_scratch_bytes(b, m);
    if (const int zrc = captra_zero_async(scratch, (size_t)sbytes, st)) return zrc;      // (a kernel, not a memset node: common.h)
    L1Params p;
    p.m2 = m2; p.fps2_idx = fps2_idx; p.new2_n3 = new2_n3; p.new2_cn = new2_cn;
    p.b = b; p.n = n; p.m = m; p.xyz_n3 = xyz_n3; p.xyz_cn = xyz_cn; p.planes = planes; p.fps_idx = fps_idx; p.new_n3 = new_n3; p.new_cn = new_cn;
    for (int s = 0; s < 3; ++s) { p.idx[s] = idx3[s]; p.r2[s] = radius3[s] * radius3[s]; }
    p.gran = reinterpret_cast<unsigned long long *>(scratch);
    p.ctl = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(scratch) + (size_t)b * m * 8);
    p.spin_limit = 2000000ull;          // 20 ms of the 100 MHz counter: a sampler takes 0.25 ms
    p.prio = g_l1_prio;
    p.dbg = g_l1_dbg;
    p.nwhole = g_l1_whole < 0 ? 0 : g_l1_whole;
    p.pair = g_l1_pair;
    p.nfine = g_l1_fine < 0 ? 0 : (g_l1_fine > m ? m : g_l1_fine) / 32 * 32;
    const int coff[3] = {0, 64, 192};
    for (int s = 0; s < 3; ++s) {
        p.net[0].img[s] = img_a3[s]; p.net[0].co_off[s] = coff[s];
        p.net[1].img[s] = cfb >= 0 ? img_b3[s] : nullptr; p.net[1].co_off[s] = coff[s];
    }
    p.net[0].feat = cfa ? feat_a : nullptr; p.net[0].out = out_a; p.net[0].out_ctotal = 320;
    p.net[1].feat = cfb > 0 ? feat_b : nullptr; p.net[1].out = out_b; p.net[1].out_ctotal = 320;
    int grid = g_l1_grid;
    if (grid <= 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        // up to 16 clouds one workgroup per CU: the launch then leaves room for whatever else runs (the other lane's networks), and
        // its own consumers, each alone on a CU, are not behind (bf16 step, two lanes of 16: 1.277 -> 1.23 ms; 512 workgroups: 1.43)
        grid = (b <= 16 ? 1 : 2) * cus;
    }
    if (grid < b + 1) grid = b + 1;
    if (cfa == 0 && cfb == 3) return l1_launch<0, 3>(p, grid, st);
    if (cfa == 3 && cfb == 0) return l1_launch<3, 0>(p, grid, st);
    if (cfa == 0 && cfb < 0) return l1_launch<0, -1>(p, grid, st);
    if (cfa == 3 && cfb < 0) return l1_launch<3, -1>(p, grid, st);
    return -2;
}

// ======================================================================================================================
// Dense CHAIN, register-resident: FP1's shared MLP + the backbone's conv1 (pointnet_utils.py:296-298, backbones.py:66-68) and,
// for CoordinateNet, both heads behind them (networks.py:29-32, 44-46) -- [c0 -> 128 -> 128 -> 128] (+ 128 -> S logits,
// 128 -> 128 -> 3P NOCS) -- as ONE launch in the bf16 mode: a wave carries 64 consecutive positions through every layer with
// the zero-swap hand-over of the SA kernels, the layers' fragment images (100-148 KB) sit in LDS for the workgroup's eight
// waves, the input is read once (fp32 channel-major rows, rounded when it becomes the first operand) and only the last
// layer(s) are stored: the 128-wide feature map as a bf16 point-major tensor (what the rotation heads' first layer reads), or
// the segmentation logits / sigmoid(NOCS) - 0.5 as fp32 (B,S,N) / (B,3P,N).  Replaces three to six launches with two to five
// round trips of (B,128,4096) tensors through HBM.
// ======================================================================================================================
namespace {

struct CbParams {
    int c0, s, no;                 // input channels, segmentation logits (0: no heads), NOCS outputs
    long long L;
    const float *x;                // (B,c0,L) fp32
    const unsigned char *img;      // fragments of every layer back to back, then the biases (32 floats per row tile)
    void *feat_pm;                 // (B,L,128) bf16 slot order, or null
    float *seg, *nocs;             // (B,s,L), (B,no,L) fp32 (heads only)
};

// one 128-wide layer from register-resident activations: RG = 2 row tiles x TN = 2 position tiles per accumulator group
template <int KST, int NOUT>
__device__ __forceinline__ void cb_layer(const unsigned char *wl, const float *bias, const u32x4 (&hin)[2][KST], u32x4 (&hout)[2][NOUT], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int tg = 0; tg < 4; tg += 2) {
        f32x16 acc[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float4 *bp = reinterpret_cast<const float4 *>(bias + 32 * (tg + r) + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = bp[2 * q];
#pragma unroll
                for (int j = 0; j < 2; ++j) { acc[r][j][4 * q + 0] = v.x; acc[r][j][4 * q + 1] = v.y; acc[r][j][4 * q + 2] = v.z; acc[r][j][4 * q + 3] = v.w; }
            }
        }
#pragma unroll
        for (int kk = 0; kk < KST; ++kk)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(wl + ((tg + r) * KST + kk) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[r][j] = sb_mfma(w, hin[j][kk], acc[r][j]);
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) sb_mid_epilogue<NOUT>(acc[r][j], tg + r, hout[j]);
    }
}

// a narrow output layer (cout <= 32: one row tile) -> fp32 rows y[(b*cout + row) * L + pos]
template <int KST>
__device__ __forceinline__ void cb_out_layer(const unsigned char *wl, const float *bias, const u32x4 (&hin)[2][KST], int cout, int act,
                                             float *y, long long L, long long pos0, int lane) {
    const int h = lane >> 5, col = lane & 31;
    f32x16 acc[2];
    const float4 *bp = reinterpret_cast<const float4 *>(bias + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = bp[2 * q];
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[j][4 * q + 0] = v.x; acc[j][4 * q + 1] = v.y; acc[j][4 * q + 2] = v.z; acc[j][4 * q + 3] = v.w; }
    }
#pragma unroll
    for (int kk = 0; kk < KST; ++kk) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(wl + kk * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = sb_mfma(w, hin[j][kk], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long c = pos0 + 32 * j + col;
        if (c >= L) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < cout) y[(size_t)row * L + c] = apply_act(acc[j][r], act);
        }
    }
}

template <int KST0, bool HEADS>
__global__ __launch_bounds__(512, 2) void chain_bf16_kernel(CbParams p) {
    constexpr int NF_TRUNK = 4 * KST0 + 32 + 32;                   // fragments of the three 128-wide layers
    constexpr int NF = NF_TRUNK + (HEADS ? 8 + 32 + 8 : 0);        // + seg (1 tile x 8), hidden (4 x 8), out (1 x 8)
    constexpr int NBT = 12 + (HEADS ? 6 : 0);                      // bias row tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *bias = reinterpret_cast<float *>(smem + NF * 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.img);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        constexpr int N16 = (NF * 1024 + NBT * 32 * 4) / 16;
        for (int e = tid; e < N16; e += 512) dst[e] = src[e];
    }
    __syncthreads();
    const int b = blockIdx.y;
    const long long pos0 = ((long long)blockIdx.x * 8 + wave) * 64;
    if (pos0 >= p.L) return;                                       // (no barrier below)
    // ---- layer 1's B operands straight from the fp32 rows: k-step kk, lane (col, h) = channels 16kk + 8h + 0..7 of its position
    const float *xb = p.x + (size_t)b * p.c0 * p.L;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.c0 * p.L * 4), 0x00020000);
    u32x4 x0[2][KST0];
    const int xrow = (int)(p.L * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        long long c = pos0 + 32 * j + col;
        if (c >= p.L) c = p.L - 1;                                 // clamped column: computed, never stored
        const int voff = (int)(((long long)(8 * h) * p.L + c) * 4); // rows >= c0 fall outside the buffer and read as 0
#pragma unroll
        for (int kk = 0; kk < KST0; ++kk) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff + (kk * 16 + i) * xrow, 0, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[j][kk][i] = sb_pack(v[2 * i], v[2 * i + 1]);
        }
    }
    u32x4 ha[2][8], hb[2][8];
    cb_layer<KST0, 8>(smem, bias, x0, ha, lane);                                                // FP1 layer 1
    cb_layer<8, 8>(smem + (4 * KST0) * 1024, bias + 128, ha, hb, lane);                         // FP1 layer 2
    cb_layer<8, 8>(smem + (4 * KST0 + 32) * 1024, bias + 256, hb, ha, lane);                    // conv1 + bn1 + ReLU
    if (p.feat_pm != nullptr) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long c = pos0 + 32 * j + col;
            if (c >= p.L) continue;
            __bf16 *yp = reinterpret_cast<__bf16 *>(p.feat_pm) + ((size_t)b * p.L + c) * 128 + 8 * h;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) *reinterpret_cast<u32x4 *>(yp + 16 * kk) = ha[j][kk];
        }
    }
    if constexpr (HEADS) {
        const unsigned char *wh = smem + NF_TRUNK * 1024;
        cb_out_layer<8>(wh, bias + 384, ha, p.s, ACT_NONE, p.seg + (size_t)b * p.s * p.L, p.L, pos0, lane);            // segmentation logits
        cb_layer<8, 8>(wh + 8 * 1024, bias + 416, ha, hb, lane);                                                        // NOCS hidden
        cb_out_layer<8>(wh + 40 * 1024, bias + 544, hb, p.no, ACT_SIGMOID_M05, p.nocs + (size_t)b * p.no * p.L, p.L, pos0, lane);
    }
}

template <int KST0, bool HEADS>
int cb_launch(int b, const CbParams &p, hipStream_t stream) {
    constexpr int NF = 4 * KST0 + 64 + (HEADS ? 48 : 0), NBT = 12 + (HEADS ? 6 : 0);
    const int lds = NF * 1024 + NBT * 32 * 4;
    auto kern = chain_bf16_kernel<KST0, HEADS>;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    CAPTRA_LAUNCH("mlp_chain3", kern, dim3((unsigned)((p.L + 511) / 512), b), dim3(512), lds, stream, p);
    return captra_last_error();
}

}  // namespace

extern "C" long long captra_chain_bf16_image_bytes(int c0, int heads) {
    if (c0 < 1 || c0 > 144) return -1;
    const long long nf = 4 * ((c0 + 15) / 16) + 64 + (heads ? 48 : 0), nbt = 12 + (heads ? 6 : 0);
    return nf * 1024 + nbt * 128;
}

// x (B,c0,L) fp32 -> relu(W3 relu(W2 relu(W1 x + b1) + b2) + b3), three 128-wide layers, in one launch; feat_pm (B,L,128) bf16
// slot order receives it when non-NULL.  heads != 0: also seg (B,s,L) = Ws feat + bs and nocs (B,no,L) = sigmoid(Wo relu(Wh feat +
// bh) + bo) - 0.5 (s, no <= 32).  img: the layers' captra_pack_dense_bf16 images back to back (layer 1 with perm = 0, the others
// with perm = 1; order: the three trunk layers, then seg, hidden, out), followed by each layer's bias as 32 floats per row tile
// (zero padded) in the same order -- captra_chain_bf16_image_bytes(c0, heads) bytes.  c0 <= 144.
extern "C" int captra_mlp_chain_bf16(int b, int c0, long long l, int heads, int s, int no, const float *x, const unsigned char *img,
                                     void *feat_pm, float *seg, float *nocs, captra_stream_t stream) {
    if (b < 0 || c0 < 1 || l < 0) return -1;
    if (c0 > 144 || (heads && (s < 1 || s > 32 || no < 1 || no > 32 || seg == nullptr || nocs == nullptr))) return -2;
    if ((long long)c0 * l * 4 >= (1ll << 31)) return -2;
    if (b == 0 || l == 0) return 0;
    CbParams p;
    p.c0 = c0; p.s = s; p.no = no; p.L = l; p.x = x; p.img = img; p.feat_pm = feat_pm; p.seg = seg; p.nocs = nocs;
    const int kst0 = (c0 + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
#define CB_CASE(K_)                                                                      \
    if (kst0 == K_) return heads ? cb_launch<K_, true>(b, p, st) : cb_launch<K_, false>(b, p, st);
    CB_CASE(9) CB_CASE(8)
#undef CB_CASE
    return -2;
}
