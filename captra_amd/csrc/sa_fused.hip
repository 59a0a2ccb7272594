// One whole set-abstraction scale in ONE kernel for gfx950:
//   ball-query neighbour list -> gather + centre-subtract + concat -> (1x1 conv + folded BN + ReLU) x 3
//   -> max over the K neighbours,
// i.e. the body of PointNetSetAbstractionMsg.forward's loop over radii (reference
// network/models/pointnet_utils.py:228-248), which the reference runs as ~12 ATen kernels over a
// materialised (B,C,S,K) tensor per scale.  Here neither the grouped tensor nor the two intermediate
// activations ever reach HBM: a workgroup owns 128 consecutive positions (= 128/K centres) and
// processes them as two 64-position sub-tiles whose gathered input X1 and activations H1, H2 live in
// LDS in the [channel][position] layout that is directly the B operand of the next layer's MFMA;
// only the pooled (C3 x centres) result is written.
//
// Operand flow (what keeps the matrix pipe fed):
//   B operand  = activations, resident in LDS, read with conflict-free ds_read_b32;
//   A operand  = weights, read STRAIGHT FROM GLOBAL MEMORY (L1/L2-resident: a scale's weights are
//                75-470 KB and every workgroup re-reads them) into the MFMA source registers, one
//                16-deep K chunk ahead of the MFMAs that consume it (buffer loads, scalar k offset).  No LDS staging of weights, hence
//                no per-chunk barriers: a sub-tile needs 4 workgroup barriers in total.
//   packed weights (captra_hip.h): zero-padded to (ceil32(cin), ceil128(cout)), so A loads need no mask.
//
// HBM traffic per scale and cloud: idx (4MK) + gathered rows + weights (both L2-resident) + 4*C3*M
// output, instead of 4*M*K*(C1 + 2*C2 + ...) bytes of activations.
//
// Arithmetic is the same k-ascending fmaf chain as pw_mlp_kernel (accumulator initialised with the
// bias, v_mfma_f32_32x32x2_f32, K never split across waves), so results are bit-identical to the
// layer-by-layer path and to the oracle.
//
// Workgroup: 512 threads = 8 waves arranged 4 (output-channel rows of 32) x 2 (position columns of 32);
// a layer is computed in output-channel tiles of 128; one 16-register accumulator per wave (the
// 32x32x2 f32 MFMA's dependent-issue latency equals its issue interval, 64 cycles), two waves per SIMD.
#include "common.h"
#include "wave_mlp.h"

namespace {


constexpr int SF_POS = 128;     // positions per workgroup (= 128/K centres), processed in sub-tiles of T = 32*WN
constexpr int SF_BK = 16;       // K chunk = 8 MFMA k-steps prefetched as one register set
constexpr int SF_BM = 128;      // output-channel tile
constexpr int SF_MAXC = 256;

struct SaParams {
    int n, m, k, cfeat;
    int c1, c2, c3;
    const float *feat;     // (B,cfeat,N) or null
    const float *xyz_cn;   // (B,3,N)
    const float *new_xyz;  // (B,M,3)
    const int *idx;        // (B,M,K)
    const float *w1, *b1, *w2, *b2, *w3, *b3;  // packed: w (ceil32(cin), ceil128(cout)), b (ceil128(cout)), zero padded
    float *out;            // (B,out_ctotal,M)
    int out_ctotal, co_off;
};

// Issue the loads of a layer's FIRST register set (16 values per lane) -- called one phase ahead of the
// layer that consumes it (before the previous layer's epilogue and the barrier), so the L2 latency of a
// layer's first weights is never exposed.  Layout of pre[] (must match sa_layer): cout <= 128: k-steps
// 0..15 of the wave's 32-column tile; cout > 128: k-steps 0..7 of tile 0 then of tile 1 (+128 columns).
template <int WN>
__device__ __forceinline__ void sa_prefetch_first(float (&pre)[16], const float *__restrict__ wt, int cin, int cout) {
    const int lane = threadIdx.x & 63;
    const int wm = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / WN;
    if (wm * 32 >= cout) return;  // wave-uniform: this wave sits the layer out
    const int ldw = (cout + 127) / 128 * 128;
    const int kp = (cin + 31) / 32 * 32;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wt, 0, kp * ldw * 4, 0x00020000);
    const int kstep_bytes = 2 * ldw * 4;
    if (cout > SF_BM) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            int col = tm * SF_BM + wm * 32 + (lane & 31);
            if (col >= ldw) col = ldw - 1;
            const int voff = (((lane >> 5) * ldw) + col) * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                pre[tm * 8 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, j * kstep_bytes, 0));
        }
    } else {
        const int voff = (((lane >> 5) * ldw) + wm * 32 + (lane & 31)) * 4;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            pre[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, j * kstep_bytes, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
}

// One layer on one T-position sub-tile: Hin [cin padded with ZERO rows][T] in LDS -> Hout / red.
//   LAST:  ReLU + max over each 32-position column block into red[row][slot]; otherwise ReLU -> Hout.
//   TM:    output-channel tiles of 128 rows computed at once (TM independent accumulator chains per wave
//          that share every B read); TM = 2 for layers wider than 128 channels (SF_MAXC = 256 = 2 tiles,
//          so a layer is always ONE pass).  A register set is 16 values: 16 k-steps (TM 1) or 8 x 2 tiles.
//   SMALL: cin <= 8 (the xyz-only SA1 input): only the first 4 k-steps of the single set are executed.
// Every set is full because the activation rows beyond cin are zero in LDS and the packed weights are
// zero there too: the inner loop is branch-free.  The A operand (weights) is fetched with buffer loads
// whose per-k offset is a SCALAR register, one set ahead of its use; the first set arrives in pre[]
// (sa_prefetch_first, issued a phase earlier) and the NEXT layer's first set is requested into pre[]
// as soon as this layer's last MFMA is issued.  The bias comes from LDS.  No barrier inside.
template <bool LAST, int WN, int TM, bool SMALL>
__device__ __forceinline__ void sa_layer(int cin, int cout, const float *__restrict__ wt, const float *bias_lds,
                                         const float *Hin, float *Hout, float *red, int red_slot, bool col_ok,
                                         float (&pre)[16], const float *__restrict__ next_wt, int next_cin, int next_cout) {
    constexpr int SF_T = 32 * WN;
    constexpr int SF_SLOTS = SF_POS / 32;
    constexpr int KS = 16 / TM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    if (wm * 32 >= cout) {  // wave-uniform: this wave's rows are all padding; it still feeds the pipeline
        sa_prefetch_first<WN>(pre, next_wt, next_cin, next_cout);
        return;
    }
    const int nsets = (cin + 2 * KS - 1) / (2 * KS);
    const int ldw = (cout + 127) / 128 * 128;
    const int kp = (cin + 31) / 32 * 32;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wt, 0, kp * ldw * 4, 0x00020000);
    const float *xrow = Hin + (lane >> 5) * SF_T + wn * 32 + (lane & 31);
    const int kstep_bytes = 2 * ldw * 4;
    const int set_bytes = KS * kstep_bytes;

    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + tm * SF_BM + wm * 32 + 4 * (lane >> 5));
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // accumulator register 4q+i holds row 8q + i (+4 for the upper half-wave)
            const float4 b4 = bp[2 * q];
            acc[tm][4 * q + 0] = b4.x; acc[tm][4 * q + 1] = b4.y; acc[tm][4 * q + 2] = b4.z; acc[tm][4 * q + 3] = b4.w;
        }
    }
    // A fragment of k-step j of set c, tile tm: W^T[c*2KS + 2j + (lane>>5)][tm*128 + wm*32 + (lane&31)]
    int voff[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int col = tm * SF_BM + wm * 32 + (lane & 31);
        if (col >= ldw) col = ldw - 1;  // (unreachable for TM = 2: ldw = 256) keep every address in bounds
        voff[tm] = (((lane >> 5) * ldw) + col) * 4;
    }
    float s0[TM][KS], s1[TM][KS], bv[KS];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int j = 0; j < KS; ++j) s0[tm][j] = pre[tm * KS + j];
#define SA_LOAD_SET(dst, set_index)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < KS; ++j)                                                                        \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) dst[tm][j] = __builtin_bit_cast(                                \
            float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[tm], (set_index) * set_bytes + j * kstep_bytes, 0));    \
    __builtin_amdgcn_sched_barrier(0);
#define SA_MFMA_SET(src, set_index, NSTEP)                                                                                 \
    {                                                                                                                      \
        const float *xr = xrow + (size_t)(set_index) * (2 * KS) * SF_T;                                                    \
        _Pragma("unroll") for (int j = 0; j < NSTEP; ++j) bv[j] = xr[j * 2 * SF_T];                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NSTEP; ++j)                                                                 \
            _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                             \
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(src[tm][j], bv[j], acc[tm], 0, 0, 0);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }
    if (SMALL) {
        SA_MFMA_SET(s0, 0, 4)
    } else {
        // Two register sets alternate; every prefetch is issued one set before its use and both halves of
        // the loop body are unconditional, so the compiler cannot sink a prefetch into a branch next to its
        // use.  An odd trailing set was prefetched by the last iteration's second slot (or arrived in pre[]
        // when there is a single set) and is consumed by the tail.
        for (int c = 0; c + 1 < nsets; c += 2) {
            SA_LOAD_SET(s1, c + 1)
            SA_MFMA_SET(s0, c, KS)
            SA_LOAD_SET(s0, (c + 2 < nsets ? c + 2 : nsets - 1))
            SA_MFMA_SET(s1, c + 1, KS)
        }
        if (nsets & 1) SA_MFMA_SET(s0, nsets - 1, KS)
    }
#undef SA_LOAD_SET
#undef SA_MFMA_SET
    sa_prefetch_first<WN>(pre, next_wt, next_cin, next_cout);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int rbase = tm * SF_BM + wm * 32 + 4 * (lane >> 5);
        if (!LAST) {
            float *hp = Hout + (size_t)rbase * SF_T + wn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float v = acc[tm][r] > 0.f ? acc[tm][r] : 0.f;
                if (rbase + ro < cout) hp[(size_t)ro * SF_T] = v;
            }
        } else {
            // max over the 32 positions of this MFMA tile: 4 DPP steps inside each row of 16 lanes, then
            // row_bcast15 folds row 0 into row 1 and row 2 into row 3 (no LDS permute); lanes 16 / 48 write
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc[tm][r] > 0.f && col_ok) ? acc[tm][r] : 0.f;
                v = row16_maxf(v);
                v = fmaxf(v, dppf_rm<0x142, 0xA>(v));
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if ((lane & 31) == 16 && row < cout) red[row * SF_SLOTS + red_slot + wn] = v;
            }
        }
    }
}

// dispatch on the layer's shape: wide layers run two output tiles at once, a tiny first layer runs 4 k-steps
template <bool LAST, int WN>
__device__ __forceinline__ void sa_layer_any(int cin, int cout, const float *wt, const float *bias_lds, const float *Hin,
                                             float *Hout, float *red, int red_slot, bool col_ok, float (&pre)[16],
                                             const float *next_wt, int next_cin, int next_cout) {
    if (cout > SF_BM) {
        sa_layer<LAST, WN, 2, false>(cin, cout, wt, bias_lds, Hin, Hout, red, red_slot, col_ok, pre, next_wt, next_cin, next_cout);
    } else if (cin <= 8) {
        sa_layer<LAST, WN, 1, true>(cin, cout, wt, bias_lds, Hin, Hout, red, red_slot, col_ok, pre, next_wt, next_cin, next_cout);
    } else {
        sa_layer<LAST, WN, 1, false>(cin, cout, wt, bias_lds, Hin, Hout, red, red_slot, col_ok, pre, next_wt, next_cin, next_cout);
    }
}

__device__ __forceinline__ int pad32(int c) { return (c + 31) & ~31; }

// zero rows [c, pad32(c)) of a [rows][64] LDS buffer (<= 31 rows)
template <int WN>
__device__ __forceinline__ void zero_pad_rows(float *buf, int c) {
    constexpr int SF_T = 32 * WN;
    const int n = (pad32(c) - c) * SF_T;
    for (int e = threadIdx.x; e < n; e += 256 * WN) buf[(size_t)c * SF_T + e] = 0.f;
}

template <int WN>
__global__ __launch_bounds__(256 * WN) __attribute__((amdgpu_waves_per_eu(4, 8))) void sa_fused_kernel(SaParams p) {
    constexpr int SF_T = 32 * WN;
    constexpr int SF_THREADS = 256 * WN;
    constexpr int SF_SUBS = SF_POS / SF_T;
    constexpr int SF_SLOTS = SF_POS / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int cin1 = p.cfeat + 3;
    const int rows_a = max(pad32(cin1), pad32(p.c2));  // region A: X1, later H2
    float *red = lds;                               // [256][4]
    float *bias_lds = red + SF_MAXC * SF_SLOTS;     // [3][256]: the three packed bias vectors
    float *RA = bias_lds + 3 * SF_MAXC;             // [rows_a][T]
    float *RB = RA + (size_t)rows_a * SF_T;         // H1 [pad32(c1)][64]

    const int b = blockIdx.y;
    const long long L = (long long)p.m * p.k;
    const long long pos0 = (long long)blockIdx.x * SF_POS;
    const int tid = threadIdx.x;
    const int wn = (tid >> 6) % WN;
    const int lane = tid & 63;
    const int gcol = tid % SF_T;   // gather: this thread's position within the sub-tile
    const int grow = tid / SF_T;   // ... and its first row (8 row groups)

    float pre[16];  // the next layer's first weight set, in flight across epilogues / barriers / the gather
    sa_prefetch_first<WN>(pre, p.w1, cin1, p.c1);
    for (int e = tid; e < 3 * SF_MAXC; e += SF_THREADS) {
        const int l = e / SF_MAXC, c = e % SF_MAXC;
        const int cl = l == 0 ? p.c1 : (l == 1 ? p.c2 : p.c3);
        const float *bl = l == 0 ? p.b1 : (l == 1 ? p.b2 : p.b3);
        bias_lds[e] = c < ((cl + 127) / 128 * 128) ? bl[c] : 0.f;
    }
    zero_pad_rows<WN>(RB, p.c1);  // H1's pad rows stay zero for the whole kernel (layers write rows < c1 only)

    for (int sub = 0; sub < SF_SUBS; ++sub) {
        const long long base = pos0 + (long long)sub * SF_T;
        // ---- gather X1 = [feat rows | xyz rows - centre] for the 64 positions of this sub-tile --------
        {
            long long pos = base + gcol;
            if (pos >= L) pos = L - 1;  // clamped column: computed, masked at the max, never stored
            const int id = p.idx[(size_t)b * L + pos];
            const float *ctr = p.new_xyz + ((size_t)b * p.m + (int)(pos / p.k)) * 3;
            const float *fb = p.feat + (size_t)b * p.cfeat * p.n + id;
            float *xcol = RA + gcol;
            int kg = grow;
            for (; kg + 56 < p.cfeat; kg += 64) {       // 8 independent loads in flight per lane
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = fb[(size_t)(kg + 8 * u) * p.n];
#pragma unroll
                for (int u = 0; u < 8; ++u) xcol[(size_t)(kg + 8 * u) * SF_T] = v[u];
            }
            for (; kg < p.cfeat; kg += 8) xcol[(size_t)kg * SF_T] = fb[(size_t)kg * p.n];
            if (grow < 3) {
                const int a = grow;
                xcol[(size_t)(p.cfeat + a) * SF_T] = p.xyz_cn[((size_t)b * 3 + a) * p.n + id] - ctr[a];
            }
            zero_pad_rows<WN>(RA, cin1);
        }
        __syncthreads();  // X1 complete (and the previous sub-tile's layer 3 is done with region A)
        sa_layer_any<false, WN>(cin1, p.c1, p.w1, bias_lds, RA, RB, red, 0, true, pre, p.w2, p.c1, p.c2);
        __syncthreads();  // H1 complete, X1 dead
        sa_layer_any<false, WN>(p.c1, p.c2, p.w2, bias_lds + SF_MAXC, RB, RA, red, 0, true, pre, p.w3, p.c2, p.c3);
        zero_pad_rows<WN>(RA, p.c2);
        __syncthreads();  // H2 complete
        const bool col_ok = (base + wn * 32 + (lane & 31)) < L;
        sa_layer_any<true, WN>(p.c2, p.c3, p.w3, bias_lds + 2 * SF_MAXC, RA, nullptr, red, sub * WN, col_ok, pre, p.w1, cin1, p.c1);
        __syncthreads();  // region A free for the next gather, red visible
    }
    // combine the 32-position maxima of each group of K positions
    const int tiles_per_group = p.k / 32;
    const int groups = SF_POS / p.k;
    for (int e = tid; e < p.c3 * groups; e += SF_THREADS) {
        const int row = e / groups, gi = e % groups;
        const long long centre = pos0 / p.k + gi;
        if (centre < p.m) {
            float v = red[row * SF_SLOTS + gi * tiles_per_group];
            for (int t = 1; t < tiles_per_group; ++t) v = fmaxf(v, red[row * SF_SLOTS + gi * tiles_per_group + t]);
            p.out[((size_t)b * p.out_ctotal + p.co_off + row) * p.m + centre] = v;
        }
    }
}

// =====================================================================================================
// Register-resident variant: sa_wave_kernel<CF, C1, C2, C3>  (channel counts are template constants)
//
// Each WAVE owns 32 positions (one 32-neighbour slice of one centre) and carries them through all three
// layers by itself.  The MFMA accumulator layout (register r of lane l = row 8(r>>2)+(r&3)+4(l>>5),
// column l&31) becomes the next layer's B operand (lane-half h of k-step j = row 2j+h) with ONE
// v_permlane32_swap per register pair, so the activations H1, H2 never leave the vector registers:
// no LDS traffic for activations, no workgroup barrier between layers, and the four waves of a workgroup
// run decoupled until the final 128-position max combine.  The gathered input is loaded from global
// memory directly in B-operand layout (lane = position, half = row parity): a 6-row SA1 input is three
// registers; the 323-row SA2 input streams through two 4-k-step register sets next to the weight sets.
// The A operand (weights) streams from L2 exactly as in sa_fused_kernel, one 16-register set ahead, with
// the hand-over between layers prefetched before the epilogue.  Same k-ascending fmaf chain: bit-identical.
// =====================================================================================================
struct SwParams {
    int b, n, m, k;
    const float *feat, *xyz_cn, *new_xyz;
    const int *idx;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;
    int out_ctotal, co_off;
    const float *v1;           // PRE kernels: (B,C1,N) = b1 + W1[feature rows] * feat per source point, else null
    unsigned long long *prof;  // debug: per-phase wave-cycle totals of a sample of waves (sa_wave_kernel), else null
    int split;                 // sa_wave_lds_kernel: a wave owns ONE 32-neighbour slice of a centre (small batches), maxima combined by atomic max
    int m0, mc;                // sa_wave_lds_kernel: centres [m0, m0 + mc) of every cloud (captra_set_centre_window; default 0, m)
    int *dyn;                  // sa_wave_lds_kernel: a zeroed counter -> centres beyond every wave's first are handed out through it
                               // (captra_sa_set_dynamic), null = static walk gid, gid + nwaves, ...
};

#define SW_TICK(slot)                                                         \
    if (CAPTRA_PROF_ON(p.prof)) {                                                  \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();        \
        if (lane == 0 && sampled) atomicAdd(p.prof + (slot), t_now - t_last); \
        t_last = t_now;                                                       \
    }

// First layer of a wide input (CF feature rows + 3 xyz rows, CF % 8 == 0, C1 <= 128): the B operand is
// gathered from global memory chunk by chunk (4 k-steps = 8 feature rows of the wave's 32 neighbours),
// double-buffered beside the matching weight chunk; all C1/32 output tiles accumulate at once so every
// gathered value is loaded exactly once.
template <int CF, int C1, int NOUT, typename Next>
__device__ __forceinline__ void sw_layer1_gather(const SwParams &p, int b, int id, const float (&ctr)[3], const float *bias_lds,
                                                 float (&hout)[NOUT], int lane, Next next) {
    constexpr int CIN = CF + 3;
    using S = SwShape<CIN, C1>;
    constexpr int NT = S::NT;
    static_assert(NT <= 4 && CF % 8 == 0, "unsupported first-layer shape");
    constexpr int NCH = CF / 8;
    const int half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, S::KP * S::LDW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)(p.feat + (size_t)b * CF * p.n), 0, CF * p.n * 4, 0x00020000);
    const int voff_w = (half * S::LDW + (lane & 31)) * 4;
    const int voff_f = (half * p.n + id) * 4;
    const int kstep_f = 2 * p.n * 4;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sw_bias_init(acc[t], bias_lds, t, lane);
    // tail operands: rows CF..CF+3 = (x, y | z, pad) relative to the centre
    float at[2][NT], bt[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            at[jj][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, voff_w, ((CF + 2 * jj) * S::LDW + 32 * t) * 4, 0));
        const int a = 2 * jj + half;
        bt[jj] = a < 3 ? p.xyz_cn[((size_t)b * 3 + a) * p.n + id] - ctr[a] : 0.f;
    }
    float A0[NT][4], A1[NT][4], B0[4], B1[4];
#define SW_LOAD_CHUNK(A, B, ch)                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                       \
        B[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, voff_f, ((ch) * 4 + j) * kstep_f, 0));   \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) A[t][j] = __builtin_bit_cast(                                       \
            float, __builtin_amdgcn_raw_buffer_load_b32(rw, voff_w, (ch) * (8 * S::LDW * 4) + (2 * j * S::LDW + 32 * t) * 4, 0)); \
    }                                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define SW_MFMA_CHUNK(A, B)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                         \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t][j], B[j], acc[t], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
    SW_LOAD_CHUNK(A0, B0, 0)
#pragma unroll 1
    for (int ch = 0; ch + 1 < NCH; ch += 2) {
        SW_LOAD_CHUNK(A1, B1, ch + 1)
        SW_MFMA_CHUNK(A0, B0)
        SW_LOAD_CHUNK(A0, B0, (ch + 2 < NCH ? ch + 2 : NCH - 1))
        SW_MFMA_CHUNK(A1, B1)
    }
    if (NCH & 1) { SW_MFMA_CHUNK(A0, B0) }
#undef SW_LOAD_CHUNK
#undef SW_MFMA_CHUNK
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[jj][t], bt[jj], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    next();
#pragma unroll
    for (int t = 0; t < NT; ++t) sw_mid_epilogue<NOUT>(acc[t], t, hout);
}

// First layer with a PRE-TRANSFORMED feature part.  The k-ascending chain of layer 1 runs over the feature rows
// first and the three relative-xyz rows last (pointnet_utils.py:234-240: cat([features, xyz])), and its first CF
// steps -- bias + sum_k W[k] feat_j[k] -- depend on the source point j only, not on the centre.  They are computed
// ONCE per source point by the dense kernel (v1 (B,C1,N) = captra_pointwise_mlp(feat, W1 rows 0..CF-1, b1, no
// activation): the very same partial fmaf chain), and here a wave just gathers them as its accumulator init and
// continues the chain with the two k-steps of the xyz rows: bit-identical to the full gather-GEMM at 1/80 of its MFMAs.
template <int CF, int C1, int NOUT, typename Next>
__device__ __forceinline__ void sw_layer1_pre(const SwParams &p, int b, int id, const float (&ctr)[3], float (&hout)[NOUT],
                                              int lane, Next next) {
    using S = SwShape<CF + 3, C1>;
    constexpr int NT = S::NT;
    static_assert(NT <= 4, "unsupported first-layer shape");
    const int half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, S::KP * S::LDW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(p.v1 + (size_t)b * C1 * p.n), 0, C1 * p.n * 4, 0x00020000);
    const int voff_w = (half * S::LDW + (lane & 31)) * 4;
    const int voff_v = (4 * half * p.n + id) * 4;
    const int row_bytes = p.n * 4;
    float at[2][NT], bt[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            at[jj][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, voff_w, ((CF + 2 * jj) * S::LDW + 32 * t) * 4, 0));
        const int a = 2 * jj + half;
        bt[jj] = a < 3 ? p.xyz_cn[((size_t)b * 3 + a) * p.n + id] - ctr[a] : 0.f;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)   // accumulator register r of this lane = row 32t + 8(r>>2) + (r&3) + 4*half, column = its neighbour
            acc[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff_v, (32 * t + 8 * (r >> 2) + (r & 3)) * row_bytes, 0));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[jj][t], bt[jj], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    next();
#pragma unroll
    for (int t = 0; t < NT; ++t) sw_mid_epilogue<NOUT>(acc[t], t, hout);
}

// live registers peak in layer 2 at about (C1 + C2)/2 activations + 64: beyond the 256 a wave gets at two
// waves per SIMD, run one wave per SIMD with the whole 512-register file instead of spilling
constexpr int sw_waves_per_simd(int c1, int c2) { return (c1 + c2) / 2 + 80 > 230 ? 1 : 2; }

template <int CF, int C1, int C2, int C3, bool PRE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (!PRE && sw_waves_per_simd(C1, C2) == 1) ? 1 : 8)))
void sa_wave_kernel(SwParams p) {
    constexpr int CIN1 = CF + 3;
    constexpr bool SMALL1 = CIN1 <= 8;
    using S1 = SwShape<CIN1, C1>;
    using S2 = SwShape<C1, C2>;
    using S3 = SwShape<C2, C3>;
    __shared__ float red[C3 * 4];
    __shared__ __attribute__((aligned(16))) float bias_lds[3 * SF_MAXC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const long long L = (long long)p.m * p.k;
    const long long pos0 = (long long)blockIdx.x * SF_POS;
    const long long wpos = pos0 + wave * 32;          // this wave's 32 positions: one slice of one centre
    const bool active = wpos < L;                     // wave-uniform (L is a multiple of 32)

    float s[2][16];
    int id = 0;
    float ctr[3] = {0.f, 0.f, 0.f};
    const bool sampled = (blockIdx.x + blockIdx.y) % 61 == 0;
    unsigned long long t_last = CAPTRA_PROF_ON(p.prof) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_first = t_last;
    const unsigned long long r_first = CAPTRA_PROF_ON(p.prof) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (active) {
        id = p.idx[(size_t)b * L + wpos + (lane & 31)];
        const float *cp = p.new_xyz + ((size_t)b * p.m + (int)(wpos / p.k)) * 3;
        ctr[0] = cp[0]; ctr[1] = cp[1]; ctr[2] = cp[2];
        if (SMALL1) sw_first_set<CIN1, C1>(s[0], p.w1, lane);
    }
    for (int e = tid; e < 3 * SF_MAXC; e += 256) {
        const int l = e / SF_MAXC, c = e % SF_MAXC;
        const int cl = l == 0 ? C1 : (l == 1 ? C2 : C3);
        const float *bl = l == 0 ? p.b1 : (l == 1 ? p.b2 : p.b3);
        bias_lds[e] = c < pad128c(cl) ? bl[c] : 0.f;
    }
    __syncthreads();
    SW_TICK(0)
    if (active) {
        float h1[S2::KST], h2[S3::KST], none[1];
        auto next2 = [&](float (&dst)[16]) { sw_first_set<C1, C2>(dst, p.w2, lane); };
        auto next3 = [&](float (&dst)[16]) { sw_first_set<C2, C3>(dst, p.w3, lane); };
        auto next_none = [&](float (&)[16]) {};
        constexpr int START2 = SMALL1 ? (S1::STEPS & 1) : 0;
        constexpr int START3 = (START2 + S2::STEPS) & 1;
        if constexpr (SMALL1) {
            float x1[S1::KST];
#pragma unroll
            for (int j = 0; j < S1::KST; ++j) {
                const int row = 2 * j + (lane >> 5);
                float v = 0.f;
                if (row < CF) v = p.feat[((size_t)b * CF + row) * p.n + id];
                else if (row < CIN1) v = p.xyz_cn[((size_t)b * 3 + (row - CF)) * p.n + id] - (row - CF == 0 ? ctr[0] : (row - CF == 1 ? ctr[1] : ctr[2]));
                x1[j] = v;
            }
            sw_layer_reg<CIN1, C1, SW_EPI_MID, 0>(p.w1, bias_lds, x1, h1, s, red, wave, lane, next2);
        } else {
            if constexpr (PRE)
                sw_layer1_pre<CF, C1>(p, b, id, ctr, h1, lane, [&]() { sw_first_set<C1, C2>(s[0], p.w2, lane); });
            else
                sw_layer1_gather<CF, C1>(p, b, id, ctr, bias_lds, h1, lane, [&]() { sw_first_set<C1, C2>(s[0], p.w2, lane); });
        }
        SW_TICK(1)
        sw_layer_reg<C1, C2, SW_EPI_MID, START2>(p.w2, bias_lds + SF_MAXC, h1, h2, s, red, wave, lane, next3);
        SW_TICK(2)
        sw_layer_reg<C2, C3, SW_EPI_MAX, START3>(p.w3, bias_lds + 2 * SF_MAXC, h2, none, s, red, wave, lane, next_none);
        SW_TICK(3)
    }
    __syncthreads();
    SW_TICK(4)
    if (CAPTRA_PROF_ON(p.prof) && lane == 0 && sampled) {
        atomicAdd(p.prof + 9, 1ull);
        atomicAdd(p.prof + 5, t_last - t_first);                                  // shader cycles of this wave's life
        atomicAdd(p.prof + 6, __builtin_amdgcn_s_memrealtime() - r_first);        // same span in 100 MHz ticks
    }
    const int tiles_per_group = p.k / 32;
    const int groups = SF_POS / p.k;
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

// =====================================================================================================
// sa_wave_lds_kernel<CF, C1, C2, C3>: the register-resident kernel for scales whose weights fit in LDS
// (the three SA1 scales: 13-74 KB).  The L1 cannot hold a scale's weights (74 KB for the widest) while 16
// waves per CU stream them at different phases, so in sa_wave_kernel ~70 % of the A-operand lines come from
// L2.  Here a persistent 8-wave workgroup stages the three weight matrices ONCE in LDS in MFMA-fragment
// order -- element ((t*KQ + q)*64 + lane)*4 + i = W'^T[2(4q+i) + (lane>>5)][32t + (lane&31)] -- so a lane
// fetches the A operands of four consecutive k-steps with one conflict-free ds_read_b128.  Each WAVE owns a
// centre and walks its K neighbours in slices of 32 with the last layer's maximum kept in registers (see the
// kernel), prefetching the next slice's neighbour ids and first-layer operand.  Everything else (gather in
// B-operand layout, activations in registers via v_permlane32_swap, integer max) is sa_wave_kernel's.  Same
// k-ascending fmaf chain: same bits.
// =====================================================================================================
constexpr int SL_WAVES = 8;
constexpr int SL_POS = SL_WAVES * 32;

template <int CIN, int COUT>
struct SlShape {
    static constexpr int KST = (CIN + 1) / 2;
    static constexpr int KQ = (KST + 3) / 4;      // quads of k-steps
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int NPASS = (NT + 1) / 2;
    static constexpr int FLOATS = NT * KQ * 256;  // fragment-ordered weights
    static constexpr int LDW = pad128c(COUT);
};

template <int CIN, int COUT>
__device__ __forceinline__ void sl_stage_weights(float *dst, const float *__restrict__ wt, int tid) {
    using S = SlShape<CIN, COUT>;
    // consecutive threads read consecutive columns of one packed row (coalesced); the LDS write is the permuted one
    for (int e = tid; e < S::FLOATS; e += SL_WAVES * 64) {
        const int l31 = e & 31, i = (e >> 5) & 3, half = (e >> 7) & 1, tq = e >> 8;
        const int q = tq % S::KQ, t = tq / S::KQ;
        const int kk = 4 * q + i;
        const float v = kk < S::KST ? wt[(size_t)(2 * kk + half) * S::LDW + 32 * t + l31] : 0.f;
        dst[((tq * 64) + half * 32 + l31) * 4 + i] = v;
    }
}

template <int CIN, int COUT, bool LAST, int NIN, int NOUT, int NZ>
__device__ __forceinline__ void sl_layer(const float *wl, const float *bias_lds, const float (&hin)[NIN], float (&hout)[NOUT],
                                         int (&zrun)[NZ], int lane) {
    using S = SlShape<CIN, COUT>;
    static_assert(NIN >= S::KST, "input operand array too small");
    const float4 *wq = reinterpret_cast<const float4 *>(wl) + lane;
#pragma unroll
    for (int ps = 0; ps < S::NPASS; ++ps) {
        f32x16 acc[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
            if (2 * ps + tm < S::NT) sw_bias_init(acc[tm], bias_lds, 2 * ps + tm, lane);
        float a[2][2][4];
#define SL_LOAD_QUAD(buf, q)                                                                   \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm) if (2 * ps + tm < S::NT) {                \
        const float4 v4 = wq[((2 * ps + tm) * S::KQ + (q)) * 64];                              \
        a[buf][tm][0] = v4.x; a[buf][tm][1] = v4.y; a[buf][tm][2] = v4.z; a[buf][tm][3] = v4.w; \
    }
        SL_LOAD_QUAD(0, 0)
#pragma unroll
        for (int q = 0; q < S::KQ; ++q) {
            if (q + 1 < S::KQ) { SL_LOAD_QUAD((q + 1) & 1, q + 1) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int kk = 4 * q + i;
                    if (kk < S::KST && 2 * ps + tm < S::NT)
                        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][tm][i], hin[kk], acc[tm], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef SL_LOAD_QUAD
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
            if (2 * ps + tm < S::NT) {
                if (LAST) sw_bfly_accumulate(acc[tm], zrun[2 * ps + tm], lane);
                else sw_mid_epilogue<NOUT>(acc[tm], 2 * ps + tm, hout);
            }
    }
}

template <int CF, int C1, int C2, int C3>
constexpr int sl_lds_floats() {
    return SlShape<CF + 3, C1>::FLOATS + SlShape<C1, C2>::FLOATS + SlShape<C2, C3>::FLOATS + pad32c(C1) + pad32c(C2) + pad32c(C3);
}

// (the kernel's body as a device function of (workgroup, workgroups): sa_wave_lds_kernel runs one scale, sa_wave_lds3_kernel the three
// scales of a level side by side in ONE launch, each on its own range of workgroups)
template <int CF, int C1, int C2, int C3>
__device__ __forceinline__ void sl_body(const SwParams &p, const int blk, const int nblk) {
    constexpr int CIN1 = CF + 3;
    static_assert(CIN1 <= 8, "LDS-weight variant is for the small-input scales");
    using S1 = SlShape<CIN1, C1>;
    using S2 = SlShape<C1, C2>;
    using S3 = SlShape<C2, C3>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl1 = lds, *wl2 = wl1 + S1::FLOATS, *wl3 = wl2 + S2::FLOATS;
    float *bias1 = wl3 + S3::FLOATS, *bias2 = bias1 + pad32c(C1), *bias3 = bias2 + pad32c(C2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // A WAVE owns a centre: it walks the centre's K neighbours in K / 32 slices of 32 (one neighbour per lane pair of
    // k-steps, as before) and keeps the running maximum of the last layer in registers (sw_bfly_accumulate), so the eight
    // waves of a workgroup share the LDS-resident weights and nothing else: no barrier after the staging one, no LDS
    // traffic for the max, no read-out pass, and the waves of a SIMD drift out of phase with each other (one's epilogue
    // under another's MFMAs) instead of meeting twice per tile.  Against the workgroup-tile form (8 waves x 32 positions,
    // maxima combined through LDS, two barriers per tile) at 32 clouds: 647 -> 635, 260 -> 237, 50.5 -> 45.7 us for the
    // three SA1 scales.  Centres are walked statically (gid, gid + nwaves, ...): every wave gets the same number.  Handing
    // centres out through a per-launch atomic counter instead was measured worse -- a centre is the unit, so the launch
    // ends up to a whole centre (4 slices, ~150 us with four waves per SIMD) late, and the atomic's return is waited for
    // at every centre end (674 / 298 / 210 us).
    const int nwaves = nblk * SL_WAVES, gid = blk * SL_WAVES + wave;
    const int ncentres = p.b * p.mc;                      // < 2^30 (launcher); centre c of the window = row crow(c) of (B, M)
    auto crow = [&](int c) { const int tb = c / p.mc; return tb * p.m + p.m0 + (c - tb * p.mc); };
    const int nslices = p.k / 32;
    auto load_ids = [&](int c, int sl) { return p.idx[(size_t)crow(c) * p.k + sl * 32 + (lane & 31)]; };
    auto load_ctr = [&](int c, float (&o)[3]) {
        const float *cp = p.new_xyz + (size_t)crow(c) * 3;
        o[0] = cp[0]; o[1] = cp[1]; o[2] = cp[2];
    };
    // first-layer B operand of a slice (k-step j = rows 2j, 2j+1: feature rows, then centre-relative xyz), gathered per lane
    auto gather_x1 = [&](int c, int id_, const float (&c_)[3], float (&x_)[S1::KST]) {
        const int tb = c / p.mc;
#pragma unroll
        for (int j = 0; j < S1::KST; ++j) {
            const int row = 2 * j + (lane >> 5);
            float v = 0.f;
            if (row < CF) v = p.feat[((size_t)tb * CF + row) * p.n + id_];
            else if (row < CIN1) v = p.xyz_cn[((size_t)tb * 3 + (row - CF)) * p.n + id_] - (row - CF == 0 ? c_[0] : (row - CF == 1 ? c_[1] : c_[2]));
            x_[j] = v;
        }
    };
    // SPLIT (small batches: fewer centres than resident waves -- a single trajectory's 512 centres would leave three quarters of
    // the chip idle while each wave walks four slices): a wave owns ONE slice, tasks t = gid, gid + nwaves, ... with centre
    // t / nslices and slice t % nslices; every task ends with its own maxima, combined across a centre's slices by an integer
    // atomic max on the pre-zeroed output (post-ReLU values are non-negative floats: they order like their bit patterns, and
    // max is exact and order-free -- the same bits as the running maximum).
    const bool split = p.split != 0;
    const int ssh = __builtin_ctz((unsigned)nslices);           // nslices is 1, 2 or 4
    int task = gid;
    int c = split ? gid >> ssh : gid, sl = split ? gid & (nslices - 1) : 0;                 // the slice being computed
    int c_next = gid + nwaves;           // the centre after c
    // DYNAMIC hand-out (p.dyn, centres of >= 2 slices): the static walk assumes every workgroup of the grid is resident.  Beside
    // another stream's one-workgroup-per-cloud samplers some are not -- they start when a resident workgroup ENDS, and the launch
    // takes twice as long (graph.py BackbonePipe).  Here every wave takes its centres from a counter: the first one before
    // anything else, each further one asked for at the current centre's FIRST slice and read at its last (that return is never
    // waited for): late workgroups find the counter exhausted and leave, the resident ones have done the work.  Which wave
    // computes a centre changes nothing in its bits.
    const bool dyn = p.dyn != nullptr && !split && nslices > 1;
    int dyn_raw = 0;
    if (dyn) {
        if (lane == 0) dyn_raw = atomicAdd(p.dyn, 1);
        c = __builtin_amdgcn_readfirstlane(dyn_raw);
    }
    int id = 0;
    float ctr[3] = {0.f, 0.f, 0.f};
    if (c < ncentres) {
        id = load_ids(c, sl);
        load_ctr(c, ctr);
    }
    sl_stage_weights<CIN1, C1>(wl1, p.w1, tid);
    sl_stage_weights<C1, C2>(wl2, p.w2, tid);
    sl_stage_weights<C2, C3>(wl3, p.w3, tid);
    for (int e = tid; e < pad32c(C1); e += SL_WAVES * 64) bias1[e] = p.b1[e];  // packed biases are zero-padded to ceil128
    for (int e = tid; e < pad32c(C2); e += SL_WAVES * 64) bias2[e] = p.b2[e];
    for (int e = tid; e < pad32c(C3); e += SL_WAVES * 64) bias3[e] = p.b3[e];
    __syncthreads();

    float x1[S1::KST];
#pragma unroll
    for (int j = 0; j < S1::KST; ++j) x1[j] = 0.f;
    if (c < ncentres) gather_x1(c, id, ctr, x1);
    int zrun[S3::NT];
    const bool sampled = blk % 16 == 0;            // phase timers (captra_sa_fused_set_prof): a sample of workgroups
    unsigned long long t_last = CAPTRA_PROF_ON(p.prof) ? __builtin_amdgcn_s_memtime() : 0ull;
    while (c < ncentres) {
        if (sl == 0 || split) {
#pragma unroll
            for (int t = 0; t < S3::NT; ++t) zrun[t] = 0;          // (0 = the ReLU)
        }
        // what comes after this slice (wave-uniform)
        const bool last_slice = split || sl + 1 == nslices;
        if (dyn) {
            if (sl == 0 && lane == 0) dyn_raw = atomicAdd(p.dyn, 1);
            if (last_slice) c_next = __builtin_amdgcn_readfirstlane(dyn_raw);
        }
        const int cn = split ? (task + nwaves) >> ssh : (last_slice ? c_next : c), sn = split ? (task + nwaves) & (nslices - 1) : (last_slice ? 0 : sl + 1);
        const bool has_next = cn < ncentres;
        int id_n = 0;
        float ctr_n[3] = {ctr[0], ctr[1], ctr[2]};
        if (has_next) {
            id_n = load_ids(cn, sn);
            if (last_slice) load_ctr(cn, ctr_n);
        }
        float h1[S2::KST], h2[S3::KST], none[1];
        float x1_n[S1::KST];
#pragma unroll
        for (int j = 0; j < S1::KST; ++j) x1_n[j] = 0.f;
        SW_TICK(0)
        sl_layer<CIN1, C1, false>(wl1, bias1, x1, h1, zrun, lane);
        SW_TICK(1)
        sl_layer<C1, C2, false>(wl2, bias2, h1, h2, zrun, lane);
        // the NEXT slice's first-layer operand: its neighbour ids were asked for two layers ago; the gather (a dependent
        // global load) runs under the widest layer
        if (has_next) gather_x1(cn, id_n, ctr_n, x1_n);
        SW_TICK(2)
        sl_layer<C2, C3, true>(wl3, bias3, h2, none, zrun, lane);
        SW_TICK(3)
        if (CAPTRA_PROF_ON(p.prof) && lane == 0 && sampled) atomicAdd(p.prof + 9, 1ull);
        if (last_slice) {
            // the centre's maxima: lane l with (l & 16) == 0 holds row 32 t + 8 ((l & 15) >> 2) + (l & 3) + 4 (l >> 5) of tile t
            const int tb = c / p.mc, centre = p.m0 + c - tb * p.mc;
            const int r = lane & 15;
            const int row0 = 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5);
            float *op = p.out + ((size_t)tb * p.out_ctotal + p.co_off + row0) * p.m + centre;
#pragma unroll
            for (int t = 0; t < S3::NT; ++t) {
                const float v = sw_bfly_finish(zrun[t]);
                if ((lane & 16) == 0 && 32 * t + row0 < C3) {
                    if (split) atomicMax(reinterpret_cast<int *>(op + (size_t)32 * t * p.m), __float_as_int(v));
                    else op[(size_t)32 * t * p.m] = v;
                }
            }
            if (!dyn) c_next += nwaves;
            ctr[0] = ctr_n[0]; ctr[1] = ctr_n[1]; ctr[2] = ctr_n[2];
            SW_TICK(4)
        }
        c = cn; sl = sn; task += nwaves;
#pragma unroll
        for (int j = 0; j < S1::KST; ++j) x1[j] = x1_n[j];
    }
}


template <int CF, int C1, int C2, int C3>
__global__ __launch_bounds__(SL_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void sa_wave_lds_kernel(SwParams p) {
    sl_body<CF, C1, C2, C3>(p, (int)blockIdx.x, (int)gridDim.x);
}

// The three small-input scales of a level in one launch (few clouds: a scale's own launch is 64-256 workgroups of a chip that holds
// 512, and the three ran one after the other): workgroups [0, g0) run scale 0, [g0, g0 + g1) scale 1, the rest scale 2 -- each
// exactly what its own launch would have run (same workgroup count, same walk), so every output bit is the same.
struct Sw3Params { SwParams s[3]; int g0, g1; };
template <int CF>
__global__ __launch_bounds__(SL_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void sa_wave_lds3_kernel(Sw3Params q) {
    const int blk = (int)blockIdx.x;
    if (blk < q.g0) sl_body<CF, 32, 32, 64>(q.s[0], blk, q.g0);
    else if (blk < q.g0 + q.g1) sl_body<CF, 64, 64, 128>(q.s[1], blk - q.g0, q.g1);
    else sl_body<CF, 64, 96, 128>(q.s[2], blk - q.g0 - q.g1, (int)gridDim.x - q.g0 - q.g1);
}

}  // namespace

// experiment knob (not part of the ABI): force the sub-tile width, 0 = heuristic
static CAPTRA_KNOB int g_sa_wn = 0;
extern "C" void captra_sa_fused_set_wn(int wn) { g_sa_wn = wn; }
static CAPTRA_KNOB int g_sa_mode = 0;  // 0 = heuristic (register-resident kernels where instantiated), 1 = always the generic LDS kernel,
                           // 2 = register-resident kernels with streamed weights only (no LDS-weight variant)
extern "C" void captra_sa_fused_set_mode(int mode) { g_sa_mode = mode; }
static CAPTRA_KNOB int g_sa_split = 1;  // a wave per slice for small batches: 0 = never, 1 = heuristic, 2 = always (tests)
extern "C" void captra_sa_fused_set_split(int v) { g_sa_split = v; }
int captra_sa_split_knob() { return g_sa_split; }
// The three small-input scales of a level as ONE launch (sa_wave_lds3_kernel): captra_sa_scales_multi (end of this file) hands
// every job's launcher a COLLECTOR on its own stack; the LDS-weight scales are recorded into it instead of launched and flushed at
// the end -- together when they are the level's three shapes in order, none with a dynamic hand-out, else one after the other as
// they would have been.  No state outside the call: any number of host threads may run it at once.
struct SlRecord {
    SwParams q;
    unsigned grid;
    int code, cf, lds;
    void (*launch)(const SwParams &, unsigned, int, hipStream_t);
};
struct SlCollect {
    SlRecord rec[3];
    int n = 0;
    alignas(16) unsigned char sp[1024];      // sa_pipe.hip's records of the second level's scales (captra_sp_collect_*)
};
template <int CF, int C1, int C2, int C3>
static void sl_launch_one(const SwParams &q, unsigned grid, int lds, hipStream_t s) {
    CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_lds_kernel<CF, C1, C2, C3>), dim3(grid), dim3(SL_WAVES * 64), lds, s, q);
}
template <int CF>
static void sl_launch_three(const SlRecord (&r)[3], hipStream_t s) {
    constexpr int lds = sl_lds_floats<CF, 64, 96, 128>() * 4;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sa_wave_lds3_kernel<CF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once.done();
    }
    Sw3Params q3;
    for (int i = 0; i < 3; ++i) q3.s[i] = r[i].q;
    q3.g0 = (int)r[0].grid; q3.g1 = (int)r[1].grid;
    CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_lds3_kernel<CF>), dim3(r[0].grid + r[1].grid + r[2].grid), dim3(SL_WAVES * 64), lds, s, q3);
}
extern void captra_sp_collect_init(void *buf, size_t bytes);      // sa_pipe.hip: the second level's two scales, recorded the same way
extern int captra_sp_collect_flush(void *buf, hipStream_t s);
static int sl_collect_flush(SlCollect &c, hipStream_t s) {
    const int n = c.n;
    c.n = 0;
    bool together = n == 3 && c.rec[0].cf == c.rec[1].cf && c.rec[1].cf == c.rec[2].cf;
    for (int i = 0; together && i < 3; ++i) together = c.rec[i].code == i && c.rec[i].q.dyn == nullptr;
    if (together && c.rec[0].cf == 0) sl_launch_three<0>(c.rec, s);
    else if (together && c.rec[0].cf == 3) sl_launch_three<3>(c.rec, s);
    else
        for (int i = 0; i < n; ++i) c.rec[i].launch(c.rec[i].q, c.rec[i].grid, c.rec[i].lds, s);
    const int err = captra_last_error();
    const int err2 = captra_sp_collect_flush(c.sp, s);
    return err != 0 ? err : err2;
}
// the slice-per-wave form's zeroed output: channels [co_off, co_off + c3), centres [m0, m0 + mc) of every cloud
static void sl_zero_window(float *out, int b, int m, int out_ctotal, int co_off, int c3, int m0, int mc, hipStream_t stream) {
    if (m0 == 0 && mc == m) {
        // (few clouds -- the only case the slice-per-wave form is chosen for -- : zeroed by a kernel per cloud, not by a memset node:
        // common.h captra_zero_async)
        if (b <= 8 && ((size_t)c3 * m * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(out + (size_t)co_off * m) & 15) == 0 && ((size_t)out_ctotal * m * 4) % 16 == 0) {
            for (int bb = 0; bb < b; ++bb) (void)captra_zero_async(out + ((size_t)bb * out_ctotal + co_off) * m, (size_t)c3 * m * 4, stream);
            return;
        }
        (void)hipMemset2DAsync(out + (size_t)co_off * m, (size_t)out_ctotal * m * 4, 0, (size_t)c3 * m * 4, b, stream);
        return;
    }
    for (int bb = 0; bb < b; ++bb)
        (void)hipMemset2DAsync(out + ((size_t)bb * out_ctotal + co_off) * m + m0, (size_t)m * 4, 0, (size_t)mc * 4, c3, stream);
}
// Dynamic centre hand-out of the persistent SA kernels (sa_wave_lds_kernel, sa_wave_pipe_kernel): the call's own device int
// (captra_launch_opts::dyn_slot), zeroed here on the launch's stream; the launch counts its centres through it.  NULL = static walk.
// A captured graph owns the slots its launches were given.
int *captra_sa_dyn_slot(const captra_launch_opts *o, hipStream_t stream) {
    if (o == nullptr || o->dyn_slot == nullptr) return nullptr;
    if (hipMemsetAsync(o->dyn_slot, 0, sizeof(int), stream) != hipSuccess) return nullptr;
    return o->dyn_slot;
}
static unsigned long long *g_sa_prof = nullptr;  // device buffer of 10 counters: sa_wave_kernel's opt-in phase timers
extern "C" void captra_sa_fused_set_prof(unsigned long long *dev_counters) { g_sa_prof = dev_counters; }
unsigned long long *captra_sa_prof_ptr() { return g_sa_prof; }

// One SA scale, fused (see include/captra_hip.h).
static int sa_scale_fused_impl(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3,
                               const float *feat, const float *xyz_cn, const float *new_xyz, const int *idx,
                               const float *w1, const float *b1, const float *w2, const float *b2,
                               const float *w3, const float *b3, float *out, int out_ctotal, int co_off,
                               const captra_launch_opts *opts, SlCollect *col, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 0 || c1 < 1 || c2 < 1 || c3 < 1) return -1;
    if (cfeat > 0 && feat == nullptr) return -1;
    if (out_ctotal < co_off + c3 || co_off < 0) return -1;
    if (k % 32 != 0 || 128 % k != 0 || c1 > SF_MAXC || c2 > SF_MAXC || c3 > SF_MAXC) return -2;
    if (b == 0 || m == 0) return 0;
    SaParams p;
    p.n = n; p.m = m; p.k = k; p.cfeat = cfeat; p.c1 = c1; p.c2 = c2; p.c3 = c3;
    p.feat = feat; p.xyz_cn = xyz_cn; p.new_xyz = new_xyz; p.idx = idx;
    p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3;
    p.out = out; p.out_ctotal = out_ctotal; p.co_off = co_off; 
    if (g_sa_mode != 1) {
        // register-resident kernels for the channel shapes of the CAPTRA backbone (network/models/pointnet_utils.py
        // PointNet2Msg config); any other shape takes the generic LDS kernel below
        SwParams q;
        q.b = b; q.n = n; q.m = m; q.k = k; q.feat = feat; q.xyz_cn = xyz_cn; q.new_xyz = new_xyz; q.idx = idx;
        q.w1 = w1; q.b1 = b1; q.w2 = w2; q.b2 = b2; q.w3 = w3; q.b3 = b3; q.out = out; q.out_ctotal = out_ctotal; q.co_off = co_off; q.v1 = nullptr; q.prof = g_sa_prof; q.split = 0; q.dyn = nullptr;
        const long long Lw = (long long)m * k;
        dim3 gridw((unsigned)((Lw + SF_POS - 1) / SF_POS), b);
        // small-input scales: persistent workgroups with the weights resident in LDS (mode 2 = streaming kernel for all)
#define SL_CASE(CF_, C1_, C2_, C3_)                                                                                   \
    if (g_sa_mode != 2 && cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_ && k % 32 == 0 && (long long)b * m < (1ll << 30)) { \
        auto kern = sa_wave_lds_kernel<CF_, C1_, C2_, C3_>;                                                            \
        constexpr int lds_bytes = sl_lds_floats<CF_, C1_, C2_, C3_>() * 4;                                             \
        static std::atomic<int> resident_of[128];              /* per device: the attribute and the occupancy are */    \
        int dev = 0;                                                                                                   \
        (void)hipGetDevice(&dev);                                                                                      \
        std::atomic<int> &resident_slot = resident_of[dev & 127];                                                      \
        int packed = resident_slot.load(std::memory_order_relaxed);      /* workgroups per CU << 16 | CUs */           \
        if (packed == 0) {                                                                                             \
            int per_cu = 0;                                                                                            \
            hipDeviceProp_t prop;                                                                                      \
            (void)hipGetDeviceProperties(&prop, dev);                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SL_WAVES * 64, lds_bytes);                \
            packed = ((per_cu > 0 ? per_cu : 1) << 16) | (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256); \
            resident_slot.store(packed, std::memory_order_relaxed);                                                    \
        }                                                                                                              \
        /* (captra_launch_opts::reserved_cus: CUs another stream's samplers hold -- every persistent workgroup must be resident) */ \
        const int cus_l = (packed & 0xffff) - captra_reserved_cus(opts) > 0 ? (packed & 0xffff) - captra_reserved_cus(opts) : 1; \
        const int resident = (packed >> 16) * cus_l;                                                                   \
        int wm0, wmc;                                                                                                  \
        (void)captra_centre_window(opts, m, &wm0, &wmc);                                                               \
        if (wmc == 0) return 0;                                                                                        \
        q.m0 = wm0; q.mc = wmc;                                                                                        \
        const long long centres = (long long)b * wmc;          /* a wave per centre: 8 centres per workgroup round */           \
        /* fewer centres than resident waves: a wave per SLICE instead (see the kernel), output pre-zeroed for the atomic max */ \
        q.split = (g_sa_split != 0 && k > 32 && (g_sa_split == 2 || centres < (long long)resident * SL_WAVES)) ? 1 : 0;    \
        const long long units = q.split ? centres * (k / 32) : centres;                                                \
        const long long wgs = (units + SL_WAVES - 1) / SL_WAVES;                                                       \
        q.b = b;                                                                                                       \
        q.dyn = (!q.split && k > 32 && wgs > 1 && col == nullptr) ? captra_sa_dyn_slot(opts, (hipStream_t)stream) : nullptr; \
        const unsigned grid_l = (unsigned)(wgs < resident ? wgs : resident);                                           \
        if (q.split && !(opts != nullptr && opts->sa_prezeroed)) sl_zero_window(out, b, m, out_ctotal, co_off, c3, wm0, wmc, (hipStream_t)stream);                  \
        if (col != nullptr && col->n < 3) {    /* captra_sa_scales_multi: recorded, launched by its flush */                \
            SlRecord &r = col->rec[col->n++];                                                                          \
            r.q = q; r.grid = grid_l; r.cf = CF_; r.lds = lds_bytes; r.launch = sl_launch_one<CF_, C1_, C2_, C3_>;      \
            r.code = (C1_ == 32 && C2_ == 32 && C3_ == 64) ? 0 : ((C1_ == 64 && C2_ == 64 && C3_ == 128) ? 1 : 2);     \
            return 0;                                                                                                  \
        }                                                                                                              \
        CAPTRA_LAUNCH("sa_scale_fused", kern, dim3(grid_l), dim3(SL_WAVES * 64), lds_bytes, (hipStream_t)stream, q);  \
        return captra_last_error();                                                                                    \
    }
        SL_CASE(0, 32, 32, 64)
        SL_CASE(0, 64, 64, 128)
        SL_CASE(0, 64, 96, 128)
        SL_CASE(3, 32, 32, 64)
        SL_CASE(3, 64, 64, 128)
        SL_CASE(3, 64, 96, 128)
#undef SL_CASE
        { int a0, ac; if (captra_centre_window(opts, m, &a0, &ac)) return -2; }   // a centre window is the LDS-weights kernels' only
#define SW_CASE(CF_, C1_, C2_, C3_)                                                                                  \
    if (cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_ && (long long)cfeat * n * 4 < (1ll << 31)) {             \
        CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_kernel<CF_, C1_, C2_, C3_>), gridw, dim3(256), 0, (hipStream_t)stream, q); \
        return captra_last_error();                                                                                  \
    }
        SW_CASE(0, 32, 32, 64)
        SW_CASE(0, 64, 64, 128)
        SW_CASE(0, 64, 96, 128)
        SW_CASE(3, 32, 32, 64)
        SW_CASE(3, 64, 64, 128)
        SW_CASE(3, 64, 96, 128)
        SW_CASE(320, 128, 128, 256)
        SW_CASE(320, 128, 196, 256)
#undef SW_CASE
    }
    const int cin1 = cfeat + 3;
    const int pa = (cin1 + 31) & ~31, pc2 = (c2 + 31) & ~31, pc1 = (c1 + 31) & ~31;
    const int rows_a = pa > pc2 ? pa : pc2;
    const long long L = (long long)m * k;
    dim3 grid((unsigned)((L + SF_POS - 1) / SF_POS), b);
    // sub-tile width: 64 positions (8 waves) when two such workgroups still fit in a CU's LDS, else 32
    // positions (4 waves): independent workgroups on a CU are what hides the gather / first-load latency
    const size_t lds64 = ((size_t)SF_MAXC * (SF_POS / 32) + 3 * SF_MAXC + (size_t)(rows_a + pc1) * 64) * sizeof(float);
    const size_t lds32 = ((size_t)SF_MAXC * (SF_POS / 32) + 3 * SF_MAXC + (size_t)(rows_a + pc1) * 32) * sizeof(float);
    int wn = (2 * lds64 <= 160 * 1024) ? 2 : 1;
    if (g_sa_wn == 1 || g_sa_wn == 2) wn = g_sa_wn;
    if ((wn == 2 ? lds64 : lds32) > 160 * 1024) return -2;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(sa_fused_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(sa_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once.done();
    }
    if (wn == 2) {
        CAPTRA_LAUNCH("sa_scale_fused", sa_fused_kernel<2>, grid, dim3(512), lds64, (hipStream_t)stream, p);
    } else {
        CAPTRA_LAUNCH("sa_scale_fused", sa_fused_kernel<1>, grid, dim3(256), lds32, (hipStream_t)stream, p);
    }
    return captra_last_error();
}

extern "C" int captra_sa_scale_fused_ex(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3,
                                        const float *feat, const float *xyz_cn, const float *new_xyz, const int *idx,
                                        const float *w1, const float *b1, const float *w2, const float *b2,
                                        const float *w3, const float *b3, float *out, int out_ctotal, int co_off,
                                        const captra_launch_opts *opts, captra_stream_t stream) {
    return sa_scale_fused_impl(b, n, m, k, cfeat, c1, c2, c3, feat, xyz_cn, new_xyz, idx, w1, b1, w2, b2, w3, b3, out, out_ctotal, co_off, opts, nullptr, stream);
}
extern "C" int captra_sa_scale_fused(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3,
                                     const float *feat, const float *xyz_cn, const float *new_xyz, const int *idx,
                                     const float *w1, const float *b1, const float *w2, const float *b2,
                                     const float *w3, const float *b3, float *out, int out_ctotal, int co_off,
                                     captra_stream_t stream) {
    return sa_scale_fused_impl(b, n, m, k, cfeat, c1, c2, c3, feat, xyz_cn, new_xyz, idx, w1, b1, w2, b2, w3, b3, out, out_ctotal, co_off, nullptr, nullptr, stream);
}

// sa_pipe.hip: captra_sa_scale_pre_pm with a collector (NULL = launch)
extern int captra_sa_scale_pre_pm_impl(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1pm,
                                       const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                                       const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                                       int out_ctotal, int co_off, const captra_launch_opts *opts, void *spbuf, captra_stream_t stream);

// A level's scales in one call (include/captra_hip.h): every job through its own launcher with a collector on THIS stack frame,
// then the flush.  A job whose shape takes no recordable kernel launches at once (in order); an error ends the call after the flush
// of what was recorded.
extern "C" int captra_sa_scales_multi(int njobs, const captra_sa_scale_job *jobs, const captra_launch_opts *opts, captra_stream_t stream) {
    if (njobs < 0 || njobs > 4 || (njobs > 0 && jobs == nullptr)) return -1;
    SlCollect col;
    captra_sp_collect_init(col.sp, sizeof(col.sp));
    captra_launch_opts o = opts != nullptr ? *opts : captra_launch_opts{};
    o.dyn_slot = nullptr;                               // (the one-launch forms walk statically; two launches cannot share a slot)
    int err = 0;
    for (int i = 0; i < njobs && err == 0; ++i) {
        const captra_sa_scale_job &j = jobs[i];
        err = j.pre ? captra_sa_scale_pre_pm_impl(j.b, j.n, j.m, j.k, j.cfeat, j.c1, j.c2, j.c3, j.feat_or_v1, j.xyz_cn, j.new_xyz, j.idx, j.w1,
                                                  j.w2, j.b2, j.w3, j.b3, j.out, j.out_ctotal, j.co_off, &o, col.sp, stream)
                    : sa_scale_fused_impl(j.b, j.n, j.m, j.k, j.cfeat, j.c1, j.c2, j.c3, j.feat_or_v1, j.xyz_cn, j.new_xyz, j.idx, j.w1, j.b1,
                                          j.w2, j.b2, j.w3, j.b3, j.out, j.out_ctotal, j.co_off, &o, &col, stream);
    }
    const int ferr = sl_collect_flush(col, (hipStream_t)stream);
    return err != 0 ? err : ferr;
}

// SA scale with a pre-transformed first layer (see sw_layer1_pre and include/captra_hip.h): v1 (B,c1,N) replaces feat.
extern "C" int captra_sa_scale_pre(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1,
                                   const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                                   const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                                   int out_ctotal, int co_off, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 1 || c1 < 1 || c2 < 1 || c3 < 1 || v1 == nullptr) return -1;
    if (out_ctotal < co_off + c3 || co_off < 0) return -1;
    if (k % 32 != 0 || 128 % k != 0) return -2;
    if ((long long)c1 * n * 4 >= (1ll << 31)) return -2;
    SwParams q;
    q.b = b; q.n = n; q.m = m; q.k = k; q.feat = nullptr; q.xyz_cn = xyz_cn; q.new_xyz = new_xyz; q.idx = idx;
    q.w1 = w1; q.b1 = b2 /* unused by the PRE kernels: any valid packed bias */; q.w2 = w2; q.b2 = b2; q.w3 = w3; q.b3 = b3;
    q.out = out; q.out_ctotal = out_ctotal; q.co_off = co_off; q.v1 = v1; q.prof = g_sa_prof; q.split = 0; q.dyn = nullptr;
    const long long Lw = (long long)m * k;
    dim3 gridw((unsigned)((Lw + SF_POS - 1) / SF_POS), b);
#define SWP_CASE(CF_, C1_, C2_, C3_)                                                                                       \
    if (cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_) {                                                             \
        if (b == 0 || m == 0) return 0;                                                                                    \
        CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_kernel<CF_, C1_, C2_, C3_, true>), gridw, dim3(256), 0, (hipStream_t)stream, q); \
        return captra_last_error();                                                                                        \
    }
    SWP_CASE(320, 128, 128, 256)
    SWP_CASE(320, 128, 196, 256)
#undef SWP_CASE
    return -2;
}
