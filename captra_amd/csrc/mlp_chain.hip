// Three dense 1x1-conv layers in ONE kernel for gfx950:  y = act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3)
// on (B, C0, L) -> (B, C3, L) -- the FP1 shared MLP followed by the backbone's conv1 + bn1 + ReLU
// (reference network/models/pointnet_utils.py:296-298 and backbones.py:66-68), which the layer kernel runs as three
// launches with two (B,128,L) round trips through HBM in between.
//
// Register-resident like sa_wave_kernel (wave_mlp.h): a wave owns 32 consecutive positions, layer 1 streams its B
// operand (two rows x 32 positions per k-step: each half-wave one 128-byte row segment) and the matching weight
// chunk from memory, double-buffered, with all C1/32 output tiles accumulating at once; its ReLU'd outputs become
// layer 2's B operands via v_permlane32_swap, likewise layer 2 -> 3; only layer 3's output is stored.  Same
// k-ascending fmaf chain per output as the layer kernel: bit-identical results.
#include "wave_mlp.h"

namespace {

struct ChainParams {
    int c0;              // (== template C0; kept for the bounds of the x buffer)
    long long L;
    const float *x;      // (B,C0,L)
    const float *w1, *b1, *w2, *b2, *w3, *b3;  // packed (captra_pack_weights)
    float *y;            // (B,C3,L)
    int act3;
};

// Layer 1: B operand from global x, all NT (<= 4) output tiles at once, chunks of 4 k-steps.
template <int C0, int C1, int NOUT, typename Next>
__device__ __forceinline__ void chain_layer1(const ChainParams &p, const float *xb, long long pos, bool col_ok, const float *bias_lds,
                                             float (&hout)[NOUT], int lane, Next next) {
    using S = SwShape<C0, C1>;
    constexpr int NT = S::NT;
    static_assert(NT <= 4, "first layer wider than 128 channels");
    constexpr int KST = S::KST;
    constexpr int NCH = KST / 4;      // full chunks
    constexpr int KT = KST - 4 * NCH;  // trailing k-steps (< 4)
    const int half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, S::KP * S::LDW * 4, 0x00020000);
    // rows >= C0 (the odd tail row) fall outside num_records and read as 0; they meet zero weight rows anyway
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)C0 * p.L * 4), 0x00020000);
    const int voff_w = (half * S::LDW + (lane & 31)) * 4;
    const long long colc = col_ok ? pos + (lane & 31) : p.L - 1;  // clamped column: computed, never stored
    // a row index beyond C0 must land beyond num_records whatever the column: add the row offset in the VECTOR offset
    const int voff_x = (int)(((long long)half * p.L + colc) * 4);
    const int kstep_x = (int)(2 * p.L * 4);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sw_bias_init(acc[t], bias_lds, t, lane);
    float A0[NT][4], A1[NT][4], B0[4], B1[4];
#define CH_LOAD(A, B, ch, nk)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < (nk); ++j) {                                                                    \
        B[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff_x + ((ch) * 4 + j) * kstep_x, 0, 0)); \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) A[t][j] = __builtin_bit_cast(                                       \
            float, __builtin_amdgcn_raw_buffer_load_b32(rw, voff_w, (ch) * (8 * S::LDW * 4) + (2 * j * S::LDW + 32 * t) * 4, 0)); \
    }                                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define CH_MFMA(A, B, nk)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < (nk); ++j)                                                                      \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t][j], B[j], acc[t], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
    // chunks 0 .. NCH-1 are full; chunk NCH (if KT > 0) has KT k-steps.  Two register sets alternate, every load one
    // chunk ahead of its use, both halves of the loop body unconditional (see sa_fused.hip on why).
    float AT[NT][4], BT[4];
    if (KT > 0) { CH_LOAD(AT, BT, NCH, KT) }
    if (NCH > 0) {
        CH_LOAD(A0, B0, 0, 4)
#pragma unroll 1
        for (int ch = 0; ch + 1 < NCH; ch += 2) {
            CH_LOAD(A1, B1, ch + 1, 4)
            CH_MFMA(A0, B0, 4)
            CH_LOAD(A0, B0, (ch + 2 < NCH ? ch + 2 : NCH - 1), 4)
            CH_MFMA(A1, B1, 4)
        }
        if (NCH & 1) { CH_MFMA(A0, B0, 4) }
    }
    if (KT > 0) { CH_MFMA(AT, BT, KT) }
#undef CH_LOAD
#undef CH_MFMA
    next();
#pragma unroll
    for (int t = 0; t < NT; ++t) sw_mid_epilogue<NOUT>(acc[t], t, hout);
}

template <int C0, int C1, int C2, int C3>
__global__ __launch_bounds__(256) void mlp_chain3_kernel(ChainParams p) {
    using S2 = SwShape<C1, C2>;
    using S3 = SwShape<C2, C3>;
    __shared__ __attribute__((aligned(16))) float bias_lds[3 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const long long pos = ((long long)blockIdx.x * 4 + wave) * 32;
    for (int e = tid; e < 3 * 256; e += 256) {
        const int l = e / 256, c = e % 256;
        const int cl = l == 0 ? C1 : (l == 1 ? C2 : C3);
        const float *bl = l == 0 ? p.b1 : (l == 1 ? p.b2 : p.b3);
        bias_lds[e] = c < pad128c(cl) ? bl[c] : 0.f;
    }
    __syncthreads();
    if (pos >= p.L) return;  // wave-uniform; no barrier below
    const bool col_ok = pos + (lane & 31) < p.L;
    float s[2][16];
    float h1[S2::KST], h2[S3::KST], none[1];
    chain_layer1<C0, C1>(p, p.x + (size_t)b * C0 * p.L, pos, col_ok, bias_lds, h1, lane,
                         [&]() { sw_first_set<C1, C2>(s[0], p.w2, lane); });
    constexpr int START3 = S2::STEPS & 1;
    sw_layer_reg<C1, C2, SW_EPI_MID, 0>(p.w2, bias_lds + 256, h1, h2, s, nullptr, wave, lane,
                                        [&](float (&dst)[16]) { sw_first_set<C2, C3>(dst, p.w3, lane); });
    SwStore st;
    st.y = p.y + (size_t)b * C3 * p.L + pos;
    st.ld = p.L;
    st.col_ok = col_ok;
    st.act = p.act3;
    sw_layer_reg<C2, C3, SW_EPI_STORE, START3>(p.w3, bias_lds + 512, h2, none, s, nullptr, wave, lane, [&](float (&)[16]) {}, st);
}

}  // namespace

// y = act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3), see include/captra_hip.h.  Returns -2 for channel shapes that are
// not instantiated (the caller then runs the three layers with captra_pointwise_mlp: same bits).
extern "C" int captra_mlp_chain3(int b, int c0, int c1, int c2, int c3, long long l, const float *x, const float *w1,
                                 const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int act3,
                                 float *y, captra_stream_t stream) {
    if (b < 0 || c0 < 1 || c1 < 1 || c2 < 1 || c3 < 1 || l < 0 || act3 < 0 || act3 > 2) return -1;
    if ((long long)c0 * l * 4 >= (1ll << 31)) return -2;  // buffer offsets are 32-bit
    ChainParams p;
    p.c0 = c0; p.L = l; p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.y = y; p.act3 = act3;
    dim3 grid((unsigned)((l + 127) / 128), b);
#define CHAIN_CASE(C0_, C1_, C2_, C3_)                                                                                   \
    if (c0 == C0_ && c1 == C1_ && c2 == C2_ && c3 == C3_) {                                                              \
        if (b == 0 || l == 0) return 0;                                                                                  \
        CAPTRA_LAUNCH("mlp_chain3", (mlp_chain3_kernel<C0_, C1_, C2_, C3_>), grid, dim3(256), 0, (hipStream_t)stream, p); \
        return captra_last_error();                                                                                      \
    }
    CHAIN_CASE(134, 128, 128, 128)  // CoordNet FP1 (xyz + xyz-as-feature + 128) -> conv1
    CHAIN_CASE(131, 128, 128, 128)  // RotationNet FP1 (xyz + 128) -> conv1
#undef CHAIN_CASE
    return -2;
}
