// bf16-input / fp32-accumulate variants of the shared-MLP kernels (BASELINE.json configs[2]: "bf16 shared-MLP on
// MFMA") for gfx950: v_mfma_f32_32x32x16_bf16.  Opt-in (cfg['mlp_dtype'] = "bf16" / fused.use_mlp_dtype); the default path and
// every parity claim of this package are exact fp32.
//
// Semantics of one layer:  y = act(b + sum_k bf16(w[k]) * bf16(x[k]))  -- weights rounded once at pack time, the input
// of every layer rounded (RNE, v_cvt_pk_bf16_f32) when it becomes an MFMA operand, products exact, accumulation and
// bias in fp32.  Activations in HBM stay fp32 (B,C,L), so every other kernel of the step is shared with the fp32 path.
//
// Operand layouts (32x32x16): A lane l = row l&31, k = 8(l>>5)..+7 -> weights are stored UNtransposed,
// Wb [ceil32(cout)][ceil32(cin)] bf16 with k contiguous: one 16-byte buffer load per lane and MFMA.  B lane l =
// column l&31, same k.  The fp32 accumulator tile turns into the next layer's B operands with one v_permlane32_swap per
// column l&31, same k.  (The SA scales of this mode live in sa_bf16.hip.)
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
    int y_pm;   // 1 = y is POINT-major (B,L,cout), cout % 4 == 0 (the SA2 scales gather it with 16-byte loads)
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
            if (col < p.L && p.y_pm) {
                float *yp = p.y + ((size_t)b * p.L + col) * p.cout + row0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (row0 + 8 * q + 3 < p.cout)
                        *reinterpret_cast<float4 *>(yp + 8 * q) = make_float4(apply_act(acc[tm][tn][4 * q], p.act), apply_act(acc[tm][tn][4 * q + 1], p.act),
                                                                              apply_act(acc[tm][tn][4 * q + 2], p.act), apply_act(acc[tm][tn][4 * q + 3], p.act));
            } else if (col < p.L) {
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

static int pw_bf16_launch(int b, int cin, int cout, long long l, const float *x, const unsigned short *wb,
                          const float *bias_packed, int act, float *y, int y_pm, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2) return -1;
    if (y_pm && cout % 4 != 0) return -2;
    if ((long long)cin * l * 4 >= (1ll << 31)) return -2;
    if (b == 0 || l == 0) return 0;
    BdParams p;
    p.cin = cin; p.cout = cout; p.kb = (cin + 31) / 32 * 32; p.L = l; p.x = x; p.wb = wb; p.bias = bias_packed; p.y = y; p.act = act; p.y_pm = y_pm;
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

extern "C" int captra_pointwise_mlp_bf16(int b, int cin, int cout, long long l, const float *x, const unsigned short *wb,
                                         const float *bias_packed, int act, float *y, captra_stream_t stream) {
    return pw_bf16_launch(b, cin, cout, l, x, wb, bias_packed, act, y, 0, stream);
}

// as captra_pointwise_mlp_bf16 with y POINT-major (B,L,cout) fp32, cout % 4 == 0
extern "C" int captra_pointwise_mlp_bf16_pm(int b, int cin, int cout, long long l, const float *x, const unsigned short *wb,
                                            const float *bias_packed, int act, float *y, captra_stream_t stream) {
    return pw_bf16_launch(b, cin, cout, l, x, wb, bias_packed, act, y, 1, stream);
}
