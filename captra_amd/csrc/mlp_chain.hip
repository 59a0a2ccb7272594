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

// CoordNet's tail in one launch: FP1 shared MLP + conv1 (-> feat, never stored) + the segmentation head (one conv) + the
// NOCS head (conv + BN + ReLU, conv, sigmoid - 0.5): six layers, two stored outputs (reference networks.py:29-32, 44-46,
// blocks.py:118-135).  Same building blocks as above; feat stays in registers as the B operand of both heads.
struct TailParams {
    long long L;
    const float *x;                                  // (B,C0,L)
    const float *w[6], *b[6];                        // fp1a, fp1b, conv1, seg, nocs hidden, nocs out (packed)
    float *seg, *nocs;                               // (B,SEG,L), (B,NOCS,L)
    int nocs_act;
};

template <int C0, int SEG, int NOCS>
__global__ __launch_bounds__(256) void coord_tail_kernel(TailParams p) {
    constexpr int C = 128;
    using SH = SwShape<C, C>;
    using SS = SwShape<C, SEG>;
    __shared__ __attribute__((aligned(16))) float bias_lds[6 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const long long pos = ((long long)blockIdx.x * 4 + wave) * 32;
    for (int e = tid; e < 6 * 256; e += 256) {
        const int l = e / 256, c = e % 256;
        bias_lds[e] = c < 128 ? p.b[l][c] : 0.f;     // packed biases: ceil128(cout) >= 128 floats each
    }
    __syncthreads();
    if (pos >= p.L) return;
    const bool col_ok = pos + (lane & 31) < p.L;
    float s[2][16];
    float h1[SH::KST], h2[SH::KST], h3[SH::KST], h4[SH::KST], none[1];
    ChainParams cp;
    cp.c0 = C0; cp.L = p.L; cp.x = p.x; cp.w1 = p.w[0];
    chain_layer1<C0, C>(cp, p.x + (size_t)b * C0 * p.L, pos, col_ok, bias_lds, h1, lane, [&]() { sw_first_set<C, C>(s[0], p.w[1], lane); });
    constexpr int P2 = 0, P3 = (P2 + SH::STEPS) & 1, P4 = (P3 + SH::STEPS) & 1, P5 = (P4 + SS::STEPS) & 1, P6 = (P5 + SH::STEPS) & 1;
    sw_layer_reg<C, C, SW_EPI_MID, P2>(p.w[1], bias_lds + 256, h1, h2, s, nullptr, wave, lane, [&](float (&d)[16]) { sw_first_set<C, C>(d, p.w[2], lane); });
    sw_layer_reg<C, C, SW_EPI_MID, P3>(p.w[2], bias_lds + 512, h2, h3, s, nullptr, wave, lane, [&](float (&d)[16]) { sw_first_set<C, SEG>(d, p.w[3], lane); });
    SwStore st;
    st.ld = p.L; st.col_ok = col_ok;
    st.y = p.seg + (size_t)b * SEG * p.L + pos; st.act = ACT_NONE;
    sw_layer_reg<C, SEG, SW_EPI_STORE, P4>(p.w[3], bias_lds + 768, h3, none, s, nullptr, wave, lane, [&](float (&d)[16]) { sw_first_set<C, C>(d, p.w[4], lane); }, st);
    sw_layer_reg<C, C, SW_EPI_MID, P5>(p.w[4], bias_lds + 1024, h3, h4, s, nullptr, wave, lane, [&](float (&d)[16]) { sw_first_set<C, NOCS>(d, p.w[5], lane); });
    st.y = p.nocs + (size_t)b * NOCS * p.L + pos; st.act = p.nocs_act;
    sw_layer_reg<C, NOCS, SW_EPI_STORE, P6>(p.w[5], bias_lds + 1280, h4, none, s, nullptr, wave, lane, [&](float (&)[16]) {}, st);
}

}  // namespace

extern "C" int captra_coord_tail(int b, int c0, int seg_dim, int nocs_dim, long long l, const float *x, const float *const *w,
                                 const float *const *bias, int nocs_act, float *seg, float *nocs, captra_stream_t stream) {
    if (b < 0 || c0 < 1 || seg_dim < 1 || nocs_dim < 1 || l < 0 || nocs_act < 0 || nocs_act > 2) return -1;
    if ((long long)c0 * l * 4 >= (1ll << 31)) return -2;
    TailParams p;
    p.L = l; p.x = x; p.seg = seg; p.nocs = nocs; p.nocs_act = nocs_act;
    for (int i = 0; i < 6; ++i) { p.w[i] = w[i]; p.b[i] = bias[i]; }
    dim3 grid((unsigned)((l + 127) / 128), b);
#define TAIL_CASE(C0_, S_, N_)                                                                                     \
    if (c0 == C0_ && seg_dim == S_ && nocs_dim == N_) {                                                            \
        if (b == 0 || l == 0) return 0;                                                                            \
        CAPTRA_LAUNCH("coord_tail", (coord_tail_kernel<C0_, S_, N_>), grid, dim3(256), 0, (hipStream_t)stream, p);  \
        return captra_last_error();                                                                                \
    }
    TAIL_CASE(134, 2, 3)    // rigid categories: 1 part + background, 3 NOCS channels
    TAIL_CASE(134, 4, 12)   // drawers: 4 parts
    TAIL_CASE(134, 3, 9)    // glasses: 3 parts
    TAIL_CASE(134, 2, 6)    // scissors / laptop: 2 parts
#undef TAIL_CASE
    return -2;
}

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
