// bf16-input / fp32-accumulate variants of the shared-MLP kernels (BASELINE.json configs[2]: "bf16 shared-MLP on
// MFMA") for gfx950: v_mfma_f32_32x32x16_bf16.  Opt-in (captra_amd.fused.MLP_DTYPE = "bf16"); the default path and
// every parity claim of this package are exact fp32.
//
// Semantics of one layer:  y = act(b + sum_k bf16(w[k]) * bf16(x[k]))  -- weights rounded once at pack time, the input
// of every layer rounded (RNE, v_cvt_pk_bf16_f32) when it becomes an MFMA operand, products exact, accumulation and
// bias in fp32.  Activations in HBM stay fp32 (B,C,L), so every other kernel of the step is shared with the fp32 path.
//
// Operand layouts (32x32x16): A lane l = row l&31, k = 8(l>>5)..+7 -> weights are stored UNtransposed,
// Wb [ceil32(cout)][ceil32(cin)] bf16 with k contiguous: one 16-byte buffer load per lane and MFMA.  B lane l =
// column l&31, same k.  The fp32 accumulator tile turns into the next layer's B operands with one v_permlane32_swap per
// register pair + packing (bw_mid_epilogue), as in wave_mlp.h: the SA scales stay register-resident.
#include "wave_mlp.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CIN, int COUT>
struct BwShape {
    static constexpr int KST = (CIN + 15) / 16;        // MFMA k-steps (16 rows each)
    static constexpr int KB = pad32c(CIN);             // bf16 elements per weight row
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int NPASS = (NT + 1) / 2;
    static constexpr int KS = 2;                       // k-steps per register set and tile: a set = 2 tiles x 2 x 4 dwords
    static constexpr int NSETS = (KST + KS - 1) / KS;
};

__global__ void pack_weights_bf16_kernel(int cin, int cout, int kb, int cp, const float *__restrict__ wt,
                                         unsigned short *__restrict__ wb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cp * kb) return;
    const int row = e / kb, k = e % kb;
    const float v = (row < cout && k < cin) ? wt[(size_t)k * cout + row] : 0.f;
    const __bf16 h = (__bf16)v;
    wb[e] = __builtin_bit_cast(unsigned short, h);
}

// ReLU, then output tile t (rows 32t..32t+31) -> B operands hout[2t], hout[2t+1] (k-steps of 16 rows)
template <int NOUT>
__device__ __forceinline__ void bw_mid_epilogue(const f32x16 &acc, int t, u32x4 (&hout)[NOUT]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        if (2 * t + jj >= NOUT) continue;
        float lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (float ReLU on purpose: the v_max_f32 feeds v_cvt_pk_bf16_f32 directly; the integer form of wave_mlp.h measured 7 % slower here)
            const float a = acc[8 * jj + i] > 0.f ? acc[8 * jj + i] : 0.f;          // rows 16jj+i   | +4 (upper half-wave)
            const float b = acc[8 * jj + 4 + i] > 0.f ? acc[8 * jj + 4 + i] : 0.f;  // rows 16jj+8+i | +4
            const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            lo[i] = __uint_as_float(p[0]);  // lower half: row 16jj+i,   upper half: row 16jj+8+i
            hi[i] = __uint_as_float(p[1]);  // lower half: row 16jj+4+i, upper half: row 16jj+12+i
        }
        u32x4 v;
        v[0] = pack_bf16(lo[0], lo[1]); v[1] = pack_bf16(lo[2], lo[3]);
        v[2] = pack_bf16(hi[0], hi[1]); v[3] = pack_bf16(hi[2], hi[3]);
        hout[2 * t + jj] = v;
    }
}

template <int CIN, int COUT, int EPI, int NIN, int NOUT>
__device__ __forceinline__ void bw_layer_reg(const unsigned short *wb, const float *bias_lds, const u32x4 (&hin)[NIN],
                                             u32x4 (&hout)[NOUT], float *red, int wave, int lane) {
    using S = BwShape<CIN, COUT>;
    static_assert(NIN >= S::KST, "input operand array too small");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wb, 0, pad32c(COUT) * S::KB * 2, 0x00020000);
    const int voff = ((lane & 31) * S::KB + 8 * (lane >> 5)) * 2;
#pragma unroll
    for (int ps = 0; ps < S::NPASS; ++ps) {
        f32x16 acc[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
            if (2 * ps + tm < S::NT) sw_bias_init(acc[tm], bias_lds, 2 * ps + tm, lane);
        u32x4 s[2][2][S::KS];
#define BW_LOAD_SET(buf, c)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < S::KS; ++j)                                                           \
        _Pragma("unroll") for (int tm = 0; tm < 2; ++tm) {                                                      \
            const int kk = (c) * S::KS + j, t = 2 * ps + tm;                                                    \
            if (kk < S::KST && t < S::NT)                                                                       \
                s[buf][tm][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (32 * t * S::KB + 16 * kk) * 2, 0); \
        }
        BW_LOAD_SET(0, 0)
#pragma unroll
        for (int c = 0; c < S::NSETS; ++c) {
            if (c + 1 < S::NSETS) { BW_LOAD_SET((c + 1) & 1, c + 1) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < S::KS; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int kk = c * S::KS + j;
                    if (kk < S::KST && 2 * ps + tm < S::NT) acc[tm] = mfma_bf16(s[c & 1][tm][j], hin[kk], acc[tm]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef BW_LOAD_SET
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
            if (2 * ps + tm < S::NT) {
                if (EPI == SW_EPI_MAX) sw_last_epilogue_bfly<COUT>(acc[tm], 2 * ps + tm, red, wave, lane);
                else bw_mid_epilogue<NOUT>(acc[tm], 2 * ps + tm, hout);
            }
    }
}

struct BsParams {
    int n, m, k;
    const float *feat;      // (B,CF,N) fp32 (small-input scales) or null
    const float *v1;        // PRE: (B,C1,N) fp32 = b1 + W1[feature rows] feat
    const float *xyz_cn, *new_xyz;
    const int *idx;
    const unsigned short *w1, *w2, *w3;  // bf16 packed; PRE: w1 holds the three xyz rows only (k = 0..2)
    const float *b1, *b2, *b3;           // fp32 packed biases
    float *out;
    int out_ctotal, co_off;
};

// One SA scale, register-resident, bf16 operands.  PRE: first layer's feature part arrives as v1 (see sa_fused.hip).
template <int CF, int C1, int C2, int C3, bool PRE>
__global__ __launch_bounds__(256) void sa_wave_bf16_kernel(BsParams p) {
    constexpr int CIN1 = PRE ? 3 : CF + 3;
    static_assert(CIN1 <= 8, "first layer: at most 8 input rows (wider inputs use the pre-transformed form)");
    using S2 = BwShape<C1, C2>;
    using S3 = BwShape<C2, C3>;
    __shared__ float red[C3 * 4];
    __shared__ __attribute__((aligned(16))) float bias_lds[3 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const long long L = (long long)p.m * p.k;
    const long long pos0 = (long long)blockIdx.x * 128;
    const long long wpos = pos0 + wave * 32;
    const bool active = wpos < L;
    int id = 0;
    float ctr[3] = {0.f, 0.f, 0.f};
    if (active) {
        id = p.idx[(size_t)b * L + wpos + (lane & 31)];
        const float *cp = p.new_xyz + ((size_t)b * p.m + (int)(wpos / p.k)) * 3;
        ctr[0] = cp[0]; ctr[1] = cp[1]; ctr[2] = cp[2];
    }
    for (int e = tid; e < 3 * 256; e += 256) {
        const int l = e / 256, c = e % 256;
        const int cl = l == 0 ? C1 : (l == 1 ? C2 : C3);
        const float *bl = l == 0 ? p.b1 : (l == 1 ? p.b2 : p.b3);
        bias_lds[e] = (c < pad128c(cl) && !(PRE && l == 0)) ? bl[c] : 0.f;   // PRE: b1 is already inside v1
    }
    __syncthreads();
    if (active) {
        // ---- layer 1: one k-step.  B operand: lower half-wave k = 0..7 = the input rows, upper half (k = 8..15) zero
        float xin[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = 0.f;
            if (!PRE && k < CF) v = p.feat[((size_t)b * CF + k) * p.n + id];
            else if (k < CIN1) {
                const int a = k - (PRE ? 0 : CF);
                v = p.xyz_cn[((size_t)b * 3 + a) * p.n + id] - (a == 0 ? ctr[0] : (a == 1 ? ctr[1] : ctr[2]));
            }
            xin[k] = (lane >> 5) ? 0.f : v;
        }
        u32x4 x1[1];
        x1[0][0] = pack_bf16(xin[0], xin[1]); x1[0][1] = pack_bf16(xin[2], xin[3]);
        x1[0][2] = pack_bf16(xin[4], xin[5]); x1[0][3] = pack_bf16(xin[6], xin[7]);
        u32x4 h1[S2::KST], h2[S3::KST], none[1];
        if constexpr (PRE) {
            // accumulator start = gathered v1 (fp32), then the xyz rows' k-step
            using S1 = BwShape<3, C1>;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, pad32c(C1) * S1::KB * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(p.v1 + (size_t)b * C1 * p.n), 0, C1 * p.n * 4, 0x00020000);
            const int voff_w = ((lane & 31) * S1::KB + 8 * (lane >> 5)) * 2;
            const int voff_v = (4 * (lane >> 5) * p.n + id) * 4;
#pragma unroll
            for (int t = 0; t < S1::NT; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff_v, (32 * t + 8 * (r >> 2) + (r & 3)) * p.n * 4, 0));
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w, (32 * t * S1::KB) * 2, 0);
                acc = mfma_bf16(a, x1[0], acc);
                bw_mid_epilogue<S2::KST>(acc, t, h1);
            }
        } else {
            bw_layer_reg<CIN1, C1, SW_EPI_MID>(p.w1, bias_lds, x1, h1, red, wave, lane);
        }
        bw_layer_reg<C1, C2, SW_EPI_MID>(p.w2, bias_lds + 256, h1, h2, red, wave, lane);
        bw_layer_reg<C2, C3, SW_EPI_MAX>(p.w3, bias_lds + 512, h2, none, red, wave, lane);
    }
    __syncthreads();
    const int tiles_per_group = p.k / 32;
    const int groups = 128 / p.k;
    for (int e = tid; e < C3 * groups; e += 256) {
        const int row = e / groups, gi = e % groups;
        const long long centre = pos0 / p.k + gi;
        if (centre < p.m) {
            float v = red[row * 4 + gi * tiles_per_group];
            for (int t = 1; t < tiles_per_group; ++t) v = fmaxf(v, red[row * 4 + gi * tiles_per_group + t]);
            p.out[((size_t)b * p.out_ctotal + p.co_off + row) * p.m + centre] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Dense layer x (B,cin,L) fp32 -> y (B,cout,L) fp32 with bf16 operands.  Both operands straight from global memory:
// A one 16-byte load per lane and MFMA; B eight row-segment loads per lane and k-step (rows 16kk + 8(l>>5) + i of the
// wave's 32 columns: each a coalesced 128-byte segment), rounded and packed in registers.  Wave tile (TM*32) x (TN*32).
// ---------------------------------------------------------------------------------------------------------------
struct BdParams {
    int cin, cout, kb;
    long long L;
    const float *x;
    const unsigned short *wb;
    const float *bias;
    float *y;
    int act;
};

template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void pw_bf16_kernel(BdParams p) {
    static_assert(WGM * WGN == 4, "4 waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int b = blockIdx.z;
    const int co0 = (blockIdx.y * WGM + wm) * TM * 32;
    const long long pos0 = ((long long)blockIdx.x * WGN + wn) * TN * 32;
    if (co0 >= p.cout) return;  // wave-uniform; no barriers in this kernel
    const int cp = (p.cout + 31) / 32 * 32;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.wb, 0, cp * p.kb * 2, 0x00020000);
    const float *xb = p.x + (size_t)b * p.cin * p.L;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.cin * p.L * 4), 0x00020000);
    int wvoff[TM], xvoff[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int row = co0 + tm * 32 + (lane & 31);
        if (row >= cp) row = cp - 1;
        wvoff[tm] = (row * p.kb + 8 * (lane >> 5)) * 2;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        long long col = pos0 + tn * 32 + (lane & 31);
        if (col >= p.L) col = p.L - 1;  // clamped column: computed, never stored
        xvoff[tn] = (int)(((long long)(8 * (lane >> 5)) * p.L + col) * 4);
    }
    const int xrow = (int)(p.L * 4);  // bytes between consecutive k rows of x
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const float *bp = p.bias + co0 + tm * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];  // packed bias (ceil128): in bounds
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = bv;
        }
    }
    const int kst = (p.cin + 15) / 16;
    u32x4 a0[TM], a1[TM];
    float b0[TN][8], b1[TN][8];
    // rows >= cin fall outside the x buffer (row offset is part of the VECTOR offset) and read as 0
#define BD_LOAD(A, Bv, kk)                                                                                            \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) A[tm] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, wvoff[tm], (kk) * 32, 0); \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) Bv[tn][i] = __builtin_bit_cast(                                   \
            float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, xvoff[tn] + ((kk) * 16 + i) * xrow, 0, 0));               \
    __builtin_amdgcn_sched_barrier(0);
#define BD_MFMA(A, Bv)                                                                                                \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                               \
        u32x4 bb;                                                                                                     \
        bb[0] = pack_bf16(Bv[tn][0], Bv[tn][1]); bb[1] = pack_bf16(Bv[tn][2], Bv[tn][3]);                             \
        bb[2] = pack_bf16(Bv[tn][4], Bv[tn][5]); bb[3] = pack_bf16(Bv[tn][6], Bv[tn][7]);                             \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = mfma_bf16(A[tm], bb, acc[tm][tn]);            \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    BD_LOAD(a0, b0, 0)
    for (int c = 0; c + 1 < kst; c += 2) {
        BD_LOAD(a1, b1, c + 1)
        BD_MFMA(a0, b0)
        BD_LOAD(a0, b0, (c + 2 < kst ? c + 2 : kst - 1))
        BD_MFMA(a1, b1)
    }
    if (kst & 1) { BD_MFMA(a0, b0) }
#undef BD_LOAD
#undef BD_MFMA
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row0 = co0 + tm * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const long long col = pos0 + tn * 32 + (lane & 31);
            if (col < p.L) {
                float *yp = p.y + ((size_t)b * p.cout + row0) * p.L + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (row0 + ro < p.cout) yp[(size_t)ro * p.L] = apply_act(acc[tm][tn][r], p.act);
                }
            }
        }
    }
}

}  // namespace

extern "C" int captra_pack_weights_bf16(int cin, int cout, const float *wt, unsigned short *wb, captra_stream_t stream) {
    if (cin < 1 || cout < 1) return -1;
    const int kb = (cin + 31) / 32 * 32, cp = (cout + 31) / 32 * 32;
    CAPTRA_LAUNCH("pack_weights", pack_weights_bf16_kernel, dim3((cp * kb + 255) / 256), dim3(256), 0, (hipStream_t)stream, cin,
                  cout, kb, cp, wt, wb);
    return captra_last_error();
}

extern "C" int captra_pointwise_mlp_bf16(int b, int cin, int cout, long long l, const float *x, const unsigned short *wb,
                                         const float *bias_packed, int act, float *y, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2) return -1;
    if ((long long)cin * l * 4 >= (1ll << 31)) return -2;
    if (b == 0 || l == 0) return 0;
    BdParams p;
    p.cin = cin; p.cout = cout; p.kb = (cin + 31) / 32 * 32; p.L = l; p.x = x; p.wb = wb; p.bias = bias_packed; p.y = y; p.act = act;
    hipStream_t s = (hipStream_t)stream;
    const long long waves22 = ((l + 63) / 64) * ((cout + 63) / 64) * b;
    if (cout >= 256 && cin >= 256 && waves22 >= 8192) {
        // wide layers: 128 x 64 wave tiles -- every converted B operand feeds four MFMAs, halving the loads per MFMA
        dim3 grid((unsigned)((l + 255) / 256), (cout + 127) / 128, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_bf16_kernel<4, 2, 1, 4>), grid, dim3(256), 0, s, p);
    } else if (cout > 64 && waves22 >= 2048) {
        dim3 grid((unsigned)((l + 127) / 128), (cout + 127) / 128, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_bf16_kernel<2, 2, 2, 2>), grid, dim3(256), 0, s, p);
    } else if (cout > 32) {
        dim3 grid((unsigned)((l + 63) / 64), (cout + 63) / 64, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_bf16_kernel<1, 1, 2, 2>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid((unsigned)((l + 255) / 256), 1, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_bf16_kernel<1, 2, 1, 4>), grid, dim3(256), 0, s, p);
    }
    return captra_last_error();
}

extern "C" int captra_sa_scale_bf16(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                                    const float *xyz_cn, const float *new_xyz, const int *idx, const unsigned short *w1,
                                    const float *b1, const unsigned short *w2, const float *b2, const unsigned short *w3,
                                    const float *b3, float *out, int out_ctotal, int co_off, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1) return -1;
    if (out_ctotal < co_off + c3 || co_off < 0) return -1;
    if (k % 32 != 0 || 128 % k != 0) return -2;
    BsParams p;
    p.n = n; p.m = m; p.k = k; p.feat = pre ? nullptr : feat_or_v1; p.v1 = pre ? feat_or_v1 : nullptr;
    p.xyz_cn = xyz_cn; p.new_xyz = new_xyz; p.idx = idx; p.w1 = w1; p.w2 = w2; p.w3 = w3; p.b1 = b1; p.b2 = b2; p.b3 = b3;
    p.out = out; p.out_ctotal = out_ctotal; p.co_off = co_off;
    const long long Lw = (long long)m * k;
    dim3 grid((unsigned)((Lw + 127) / 128), b);
#define BS_CASE(CF_, C1_, C2_, C3_, PRE_)                                                                                  \
    if (cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_ && (pre != 0) == PRE_) {                                       \
        if (b == 0 || m == 0) return 0;                                                                                    \
        CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_bf16_kernel<CF_, C1_, C2_, C3_, PRE_>), grid, dim3(256), 0, (hipStream_t)stream, p); \
        return captra_last_error();                                                                                        \
    }
    BS_CASE(0, 32, 32, 64, false) BS_CASE(0, 64, 64, 128, false) BS_CASE(0, 64, 96, 128, false)
    BS_CASE(3, 32, 32, 64, false) BS_CASE(3, 64, 64, 128, false) BS_CASE(3, 64, 96, 128, false)
    BS_CASE(320, 128, 128, 256, true) BS_CASE(320, 128, 196, 256, true)
#undef BS_CASE
    return -2;
}
