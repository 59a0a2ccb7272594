// Shared-MLP layers (1x1 conv + folded BatchNorm + activation) as exact-fp32 MFMA GEMMs on gfx950.
//
// Replaces the Conv2d/Conv1d(1x1) -> BatchNorm -> ReLU runs of the reference
// (network/models/pointnet_utils.py:242-246, 296-298, 336-340; backbones.py:68), which the
// reference executes as separate ATen kernels over a MATERIALISED grouped tensor.
//
// One kernel template, two operand-load prologues and two epilogues:
//   PRO_PLAIN  x (B,cin,L) dense
//   PRO_GROUP  x gathered on the fly through the ball-query index list, centre subtracted,
//              [feat, xyz] concatenated — the (B,cin,M,K) grouped tensor is never written
//              (replaces group_operation + '-=' + cat, pointnet_utils.py:234-240)
//   EPI_STORE  y (B,cout,L) = act(acc)
//   EPI_MAXK   y[b][co][m] = max over the K neighbours of relu(acc) — the last layer's (B,cout,M,K)
//              activation is never written (replaces torch.max(-1), pointnet_utils.py:246)
//
// GEMM mapping: rows = output channels (A = W^T tile [k][co] in LDS), cols = positions
// (B = X tile [k][pos] in LDS), v_mfma_f32_32x32x2_f32.  Both fragments are read with
// conflict-free ds_read_b32 (lane l -> row k0+(l>>5), column base+(l&31)).
// Arithmetic contract (include/captra_hip.h): acc = bias; acc = fmaf(W[co][k], x[k][pos], acc)
// for k ascending — exactly what a chain of 32x32x2 f32 MFMAs computes — then the activation.
// K is never split across waves, so the result is bit-identical to the oracle's fmaf loop.
//
// Weights arrive PACKED (captra_hip.h "packed weights"): W^T zero-padded to (ceil32(cin), ceil128(cout))
// and the bias to ceil128(cout).  Every weight/bias access is therefore in bounds and needs no mask,
// X rows beyond cin are clamped to the last real row (they meet zero weights), positions beyond L
// are clamped (their columns are never stored): the staging code is pointer bumps + 16-byte loads
// with no predicates, which is what keeps the VALU out of the MFMA pipe's way.
//
// Pipeline: register-staged prefetch (global loads of chunk c+1 are issued before the MFMAs of
// chunk c and written to LDS after them), 2 workgroups per CU.
#include "common.h"
#include <atomic>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PW_THREADS = 256;
constexpr int PW_BK = 32;

enum { PRO_PLAIN = 0, PRO_GROUP = 1 };
enum { EPI_STORE = 0, EPI_MAXK = 1 };

struct PwParams {
    int cin, cout, ldw;   // ldw = ceil128(cout): row stride of the packed weights
    long long L;          // positions per cloud (= M*K for grouped layers)
    const float *x;       // PRO_PLAIN: (B,cin,L)
    const float *wt;      // packed (ceil32(cin), ldw)
    const float *bias;    // packed (ldw)
    long long bias_bs;    // floats between two clouds' biases (0: one bias for all; captra_pointwise_mlp_cb)
    float *y;
    int act;
    // PRO_GROUP
    int n, m, k, cfeat;
    const float *feat;    // (B,cfeat,N) or null
    const float *xyz_cn;  // (B,3,N)
    const float *new_xyz; // (B,M,3)
    const int *idx;       // (B,M,K)
    // EPI_MAXK
    int y_ctotal, co_off; // y is (B,y_ctotal,M), this layer writes channels [co_off, co_off+cout)
    // GroupNorm fusion of the direct kernel (captra_pointwise_mlp_gn)
    const float *ab_in;   // (B,cin,2) or null: the input is relu(a*x + b) per (cloud, input channel)
    float *stats_out;     // (B,cout,T,2) or null: per output row and 64-column tile, (sum, sum of squares) of y
    int stats_t;          // T
    int y_pm;             // direct kernel only: 1 = y is POINT-major (B,L,cout), cout % 4 == 0 (captra_pointwise_mlp_pm)
    // SRC2 (captra_pointwise_mlp2): input channels >= csplit come from x2 (B,cin - csplit,L2), L2 = L or 1 (one vector per
    // cloud, read for every position); both tensors are addressed from ONE buffer descriptor based at the lower of the two
    const float *x2;
    int csplit, x2_bcast;
    int dbg;              // CAPTRA_ABLATIONS builds only (timing ablations of the direct kernel, results wrong; != 0 takes the general epilogue): 1 = no y stores, 2 = no statistics, 4 = every workgroup reads the first columns, 8 = one statistics store per row tile
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// max over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}

// BM x BN output tile per workgroup, 4 waves arranged WGM x WGN, wave tile (TM*32) x (TN*32).
template <int BM, int BN, int WGM, int WGN, int PRO, int EPI, bool VECX>
__global__ __launch_bounds__(PW_THREADS) void pw_mlp_kernel(PwParams p) {
    constexpr int TM = BM / WGM / 32;
    constexpr int TN = BN / WGN / 32;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int W4 = PW_BK * BM / 4 / PW_THREADS;  // float4 of W staged per thread per chunk
    constexpr int X4 = PW_BK * BN / 4 / PW_THREADS;  // float4 of X staged per thread per chunk (vector form)
    constexpr int X1 = PW_BK * BN / PW_THREADS;      // floats of X staged per thread per chunk (scalar forms)
    static_assert(W4 >= 1, "W tile too small for 16-byte staging");

    __shared__ __attribute__((aligned(16))) float Ws[PW_BK * BM];
    __shared__ __attribute__((aligned(16))) float Xs[PW_BK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * BM;
    const long long pos0 = (long long)blockIdx.x * BN;

    // ---- staging pointers, set up once; per chunk they advance by a constant stride ----------------
    // W: float4 e = tid + i*256 -> row e / (BM/4), col4 e % (BM/4); always in bounds (packed weights)
    const float *wptr[W4];
#pragma unroll
    for (int i = 0; i < W4; ++i) {
        const int e = tid + i * PW_THREADS;
        wptr[i] = p.wt + (size_t)(e / (BM / 4)) * p.ldw + co0 + (e % (BM / 4)) * 4;
    }
    const size_t wstep = (size_t)PW_BK * p.ldw;
    float4 wreg[W4];

    // X (plain): rows clamped to cin-1 (they meet zero weight rows), columns clamped into [0, L)
    float4 xreg4[VECX ? X4 : 1];
    float xreg1[VECX ? 1 : X1];
    int xrow_local[VECX ? X4 : X1];
    long long xcol[VECX ? X4 : X1];
    const float *xbase = p.x + (size_t)b * p.cin * p.L;

    // PRO_GROUP: BN is a multiple of 256 or divides it, so a thread's column is fixed
    int g_id = 0;
    float g_c[3] = {0.f, 0.f, 0.f};
    if (PRO == PRO_GROUP) {
        long long pos = pos0 + tid % BN;
        if (pos >= p.L) pos = p.L - 1;  // clamped column: computed, never stored
        g_id = p.idx[(size_t)b * p.L + pos];
        const float *c = p.new_xyz + ((size_t)b * p.m + (int)(pos / p.k)) * 3;
        g_c[0] = c[0];
        g_c[1] = c[1];
        g_c[2] = c[2];
    } else if (VECX) {
#pragma unroll
        for (int i = 0; i < X4; ++i) {
            const int e = tid + i * PW_THREADS;
            xrow_local[i] = e / (BN / 4);
            const long long pos = pos0 + (e % (BN / 4)) * 4;
            xcol[i] = pos < p.L ? pos : (p.L - 4);  // L % 4 == 0: a float4 is all-in or all-out
        }
    } else {
#pragma unroll
        for (int i = 0; i < X1; ++i) {
            const int e = tid + i * PW_THREADS;
            xrow_local[i] = e / BN;
            const long long pos = pos0 + e % BN;
            xcol[i] = pos < p.L ? pos : (p.L - 1);
        }
    }

    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < W4; ++i) {
            wreg[i] = *reinterpret_cast<const float4 *>(wptr[i]);
            wptr[i] += wstep;
        }
        if (PRO == PRO_PLAIN) {
            if (VECX) {
#pragma unroll
                for (int i = 0; i < X4; ++i) {
                    const int kg = min(kc + xrow_local[i], p.cin - 1);
                    xreg4[i] = *reinterpret_cast<const float4 *>(xbase + (size_t)kg * p.L + xcol[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < X1; ++i) {
                    const int kg = min(kc + xrow_local[i], p.cin - 1);
                    xreg1[i] = xbase[(size_t)kg * p.L + xcol[i]];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < X1; ++i) {
                const int kg = kc + (tid + i * PW_THREADS) / BN;
                // rows >= cfeat+3 are clamped onto the last xyz row: finite values against zero weights
                const bool is_feat = kg < p.cfeat;
                const int a = min(max(kg - p.cfeat, 0), 2);
                const float *rowp = is_feat ? (p.feat + ((size_t)b * p.cfeat + kg) * p.n) : (p.xyz_cn + ((size_t)b * 3 + a) * p.n);
                const float raw = rowp[g_id];
                const float ctr = a == 0 ? g_c[0] : (a == 1 ? g_c[1] : g_c[2]);
                xreg1[i] = is_feat ? raw : (raw - ctr);
            }
        }
    };

    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < W4; ++i) *reinterpret_cast<float4 *>(Ws + (size_t)(tid + i * PW_THREADS) * 4) = wreg[i];
        if (PRO == PRO_PLAIN && VECX) {
#pragma unroll
            for (int i = 0; i < X4; ++i) *reinterpret_cast<float4 *>(Xs + (size_t)(tid + i * PW_THREADS) * 4) = xreg4[i];
        } else {
#pragma unroll
            for (int i = 0; i < X1; ++i) Xs[tid + i * PW_THREADS] = xreg1[i];
        }
    };

    // ---- accumulators start at the bias (the fmaf chain's C input) ------------------------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const float *bp = p.bias + (size_t)blockIdx.z * p.bias_bs + co0 + (wm * TM + tm) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];  // packed bias: in bounds, zero beyond cout
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = bv;
        }
    }

    const int nchunks = (p.cin + PW_BK - 1) / PW_BK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // previous chunk's MFMAs have consumed the LDS tiles
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk((c + 1) * PW_BK);  // in flight during the MFMAs below
        const int kleft = p.cin - c * PW_BK;
        const int ksteps = kleft >= PW_BK ? PW_BK : ((kleft + 1) & ~1);
        const float *wrow = Ws + (lane >> 5) * BM + wm * TM * 32 + (lane & 31);
        const float *xrow = Xs + (lane >> 5) * BN + wn * TN * 32 + (lane & 31);
        for (int kk = 0; kk < ksteps; kk += 2) {
            float a[TM], bb[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = wrow[kk * BM + tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bb[tn] = xrow[kk * BN + tn * 32];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------
    if (EPI == EPI_STORE) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row0 = co0 + (wm * TM + tm) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const long long col = pos0 + (wn * TN + tn) * 32 + (lane & 31);
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
    } else {
        // max over groups of K consecutive positions; K % 32 == 0 and BN % K == 0 (host checks).
        // Stage 1: per 32-position MFMA tile, reduce over the 32 lanes that hold one output row.
        // Stage 2: combine the K/32 tiles of a group through LDS (reusing the X staging buffer).
        __syncthreads();  // everyone is done reading Xs
        float *red = Xs;  // [BM][BN/32]
        constexpr int NT = BN / 32;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const bool col_ok = pos0 + (wn * TN + tn) * 32 + (lane & 31) < p.L;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[tm][tn][r];
                    v = (v > 0.f && col_ok) ? v : 0.f;  // ReLU (the only activation followed by a max in the path)
                    v = row16_max(v);
                    v = fmaxf(v, __shfl_xor(v, 16, 64));
                    if ((lane & 31) == 0) {
                        const int rl = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        red[rl * NT + wn * TN + tn] = v;
                    }
                }
            }
        __syncthreads();
        const int tiles_per_group = p.k / 32;
        const int groups = BN / p.k;
        for (int e = tid; e < BM * groups; e += PW_THREADS) {
            const int rl = e / groups, g = e % groups;
            const int row = co0 + rl;
            const long long centre = pos0 / p.k + g;
            if (row < p.cout && centre < p.m) {
                float v = red[rl * NT + g * tiles_per_group];
                for (int t = 1; t < tiles_per_group; ++t) v = fmaxf(v, red[rl * NT + g * tiles_per_group + t]);
                p.y[((size_t)b * p.y_ctotal + p.co_off + row) * p.m + centre] = v;
            }
        }
    }
}

template <int PRO, int EPI, bool VECX>
int launch_pw(int b, const PwParams &p, hipStream_t s, const char *name) {
    // tile choice by output-channel count: wide tiles for wide layers, BN=256 for narrow ones
    if (p.cout > 64) {
        dim3 grid((unsigned)((p.L + 127) / 128), (p.cout + 127) / 128, b);
        CAPTRA_LAUNCH(name, (pw_mlp_kernel<128, 128, 2, 2, PRO, EPI, VECX>), grid, dim3(PW_THREADS), 0, s, p);
    } else if (p.cout > 32) {
        dim3 grid((unsigned)((p.L + 255) / 256), (p.cout + 63) / 64, b);
        CAPTRA_LAUNCH(name, (pw_mlp_kernel<64, 256, 1, 4, PRO, EPI, VECX>), grid, dim3(PW_THREADS), 0, s, p);
    } else {
        dim3 grid((unsigned)((p.L + 255) / 256), 1, b);
        CAPTRA_LAUNCH(name, (pw_mlp_kernel<32, 256, 1, 4, PRO, EPI, VECX>), grid, dim3(PW_THREADS), 0, s, p);
    }
    return captra_last_error();
}


// ---------------------------------------------------------------------------------------------
// Direct-operand variant for dense layers (x (B,cin,L) -> y (B,cout,L)): NO LDS and NO barriers.
// Both MFMA operands are fetched with buffer loads straight into the MFMA source registers, one
// register set (8 k-steps) ahead of the MFMAs that consume it:
//   A = packed weights (L1/L2-resident), scalar k offset;
//   B = activations: lane l reads x[k0 + 2j + (l>>5)][pos + (l&31)] — each half-wave one 128-byte row
//       segment; rows beyond cin fall outside the buffer's num_records and read as 0 (and meet zero
//       weight rows anyway), columns beyond L are clamped (computed, never stored).
// A wave owns a (TM*32) x (TN*32) tile with TM*TN independent accumulators; the four waves of a
// workgroup (WGM x WGN) share operand rows through L1.  Same k-ascending fmaf chain -> same bits.
// ---------------------------------------------------------------------------------------------
// AFF: the layer's input is the previous layer's raw output under that layer's GroupNorm + ReLU, applied on the fly
//      to every B operand as relu(a*x + b) with per-(cloud, channel) coefficients (captra_gn_finalize);
// ST:  the epilogue also reduces this layer's raw output to per-row, per-64-column-tile (sum, sum of squares)
//      partials -- fixed DPP order, no atomics -- from which captra_gn_finalize derives the next coefficients.
// Together they remove the separate GroupNorm pass (one read and one write of the activation tensor per layer).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {  // 0 is the identity: lanes without a source / masked rows add 0
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
// sum over each half-wave (32 lanes): valid in lanes 16..31 (lower half) and 48..63 (upper half)
__device__ __forceinline__ float half_wave_sum(float v) {
    v = dpp_add_f32<0xB1, 0xF>(v);
    v = dpp_add_f32<0x4E, 0xF>(v);
    v = dpp_add_f32<0x141, 0xF>(v);
    v = dpp_add_f32<0x140, 0xF>(v);   // every lane of a row of 16 holds the row sum
    v = dpp_add_f32<0x142, 0xA>(v);   // row_bcast15: rows 1 and 3 add rows 0 and 2
    return v;
}

// Transposing reduction of the statistics epilogue: 16 per-row values per lane -> ONE value per lane, the total over the half-wave's
// 32 lanes of the row the lane stands for.  Each step halves the registers: a lane keeps the first of a register pair where its lane
// bit is 0 and the second where it is 1, and adds its partner's copy of the one it keeps.  Bits 2 and 3 pick whole banks of four
// lanes, so the keep / send choice is the DPP move's bank mask (no select); bits 0 and 1 take two v_cndmask.  15 exchanges of three
// instructions instead of 16 five-step butterflies, and one store per lane at the end instead of two two-lane stores per row.
template <int CTRL_LO, int CTRL_HI, int BANK_LO, int BANK_HI>
__device__ __forceinline__ float xchg_add_banked(float a, float b) {
    const int x = __builtin_amdgcn_update_dpp(__float_as_int(b), __float_as_int(a), CTRL_LO, 0xF, BANK_LO, false);   // bit 0 lanes: the partner's a
    const int y = __builtin_amdgcn_update_dpp(__float_as_int(a), __float_as_int(b), CTRL_HI, 0xF, BANK_HI, false);   // bit 1 lanes: the partner's b
    return __int_as_float(x) + __int_as_float(y);
}
template <int CTRL>
__device__ __forceinline__ float xchg_add_select(float a, float b, bool bit) {
    const float keep = bit ? b : a, send = bit ? a : b;
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, true));
}
// v[16] -> the half-wave total of register rr(lane) = ((lane >> 1) & 1) + 2 (lane & 1) + 4 ((lane >> 3) & 1) + 8 ((lane >> 2) & 1)
__device__ __forceinline__ float half_wave_transpose_sum(const float (&v)[16], int lane) {
    float w[8], u[4], t[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = xchg_add_banked<0x104, 0x114, 0x5, 0xA>(v[k], v[k + 8]);      // lane ^ 4: row_shl:4 / row_shr:4
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = xchg_add_banked<0x128, 0x128, 0x3, 0xC>(w[k], w[k + 4]);      // lane ^ 8: row_ror:8
#pragma unroll
    for (int k = 0; k < 2; ++k) t[k] = xchg_add_select<0xB1>(u[k], u[k + 2], (lane & 1) != 0);       // lane ^ 1: quad_perm [1,0,3,2]
    float s = xchg_add_select<0x4E>(t[0], t[1], (lane & 2) != 0);                                     // lane ^ 2: quad_perm [2,3,0,1]
    return s + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s), 0x401F));                // lane ^ 16 (swizzle within 32 lanes)
}

// PAIR (TN == 2, L even): the wave's 64 columns are dealt to its two column tiles ALTERNATELY (tile tn holds columns
//      pos0 + 2i + tn), so one 8-byte load per lane and k-step feeds both tiles' B operands and one 8-byte store writes
//      both tiles' outputs of a row -- half the vector-memory instructions of the B side.  Which column a lane's
//      accumulator stands for changes, the per-element k-ascending chain does not: same bits per output element.
template <int ACT>
__device__ __forceinline__ float apply_act_c(float v) {
    if (ACT == ACT_RELU) return relu_bits(v);
    if (ACT == ACT_SIGMOID_M05) return 1.0f / (1.0f + expf(-v)) - 0.5f;
    return v;
}

template <int TM, int TN, int WGM, int WGN, bool AFF = false, bool ST = false, bool PAIR = false, bool SRC2 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void pw_direct_kernel(PwParams p) {
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(!PAIR || TN == 2, "paired column tiles");
    static_assert(!SRC2 || (!PAIR && !AFF), "two-source input: plain layers only");
    constexpr int KS = (AFF || (TM == 2 && TN == 2)) ? 4 : 8;  // k-steps per register set
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int dbg = CAPTRA_ABLATIONS ? p.dbg : 0;
    const int b = blockIdx.z;
    const int co0 = (blockIdx.y * WGM + wm) * TM * 32;
    const long long pos0 = ((long long)blockIdx.x * WGN + wn) * TN * 32;
    if (co0 >= p.cout) return;  // wave-uniform; no barriers in this kernel
    // A operands from the FRAGMENT image of the packed buffer (behind its row-major image): one 16-byte load per lane holds
    // four consecutive k-steps of an output tile -- a quarter of the vector-memory instructions of dword loads, which at one
    // load per MFMA kept the CU's address unit as busy as its four matrix pipes
    const int kp = (p.cin + 31) / 32 * 32;
    const int kq = ((p.cin + 1) / 2 + 3) / 4;                  // quads of k-steps per output tile
    const int ntile = (p.cout + 31) / 32;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.wt + (size_t)kp * p.ldw), 0, ntile * kq * 1024, 0x00020000);
    const float *xb = p.x + (size_t)b * (SRC2 ? p.csplit : p.cin) * p.L;
    // SRC2: the concat [x; x2] is never built -- row r of the operand is row r of x (r < csplit) or row r - csplit of x2; one
    // descriptor based at the lower of the two per-cloud blocks, a per-lane byte offset per k-step (the launcher checked that both
    // blocks lie within 2^31 bytes of it)
    const float *x2b = SRC2 ? p.x2 + (size_t)b * (p.cin - p.csplit) * (p.x2_bcast ? 1 : p.L) : nullptr;
    const float *xbase = SRC2 && x2b < xb ? x2b : xb;
    const int off1 = SRC2 ? (int)((const char *)xb - (const char *)xbase) : 0, off2 = SRC2 ? (int)((const char *)x2b - (const char *)xbase) : 0;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xbase, 0, SRC2 ? 0x7ffffff0 : (int)((long long)p.cin * p.L * 4), 0x00020000);
    const int xstep = (int)(2 * p.L * 4);     // bytes per k-step in X
    int xcol4[TN];                            // SRC2: byte offset of this lane's column in a row
    int wvoff[TM], xvoff[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int tile = co0 / 32 + tm;
        if (tile >= ntile) tile = ntile - 1;                    // rows beyond cout: any valid tile (their results are not stored)
        wvoff[tm] = tile * kq * 1024 + lane * 16;
    }
    // column of this lane in column tile tn
    auto col_of = [&](int tn) { return PAIR ? pos0 + 2 * (lane & 31) + tn : pos0 + tn * 32 + (lane & 31); };
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        long long col = col_of(tn);
        if (col >= p.L) col = PAIR ? p.L - 2 + tn : p.L - 1;   // (PAIR: L and the pair's first column are even)
        if (dbg & 4) col -= pos0;
        xvoff[tn] = (int)(((long long)(lane >> 5) * p.L + col) * 4);
        xcol4[tn] = (int)(col * 4);
    }
    // SRC2: byte offset of row 2 ks + (lane >> 5) of the virtual concat, column tile tn (rows beyond cin: the last row, their weights are zero)
    auto src2_off = [&](int ks, int tn) {
        const int r = 2 * ks + (lane >> 5);
        if (r >= p.cin) return 0x7ffffffc;      // beyond the descriptor: reads as 0, like the one-tensor kernel's rows >= cin (a set may run
                                                // past the layer's last k-step quad, where the A fragments are another tile's weights)
        return r < p.csplit ? off1 + r * (int)(p.L * 4) + xcol4[tn]
                            : off2 + (r - p.csplit) * (p.x2_bcast ? 4 : (int)(p.L * 4)) + (p.x2_bcast ? 0 : xcol4[tn]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const float *bp = p.bias + (size_t)b * p.bias_bs + co0 + tm * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = bv;
        }
    }

    float a0[TM][KS], a1[TM][KS], b0[TN][KS], b1[TN][KS];
    // GroupNorm coefficients (a, b) of this lane half's row of every k-step of a set: one 8-byte load per k-step (two
    // addresses per wave).  Through the scalar cache instead (the rows are wave-uniform) the pair costs four v_mov + two
    // v_cndmask per k-step in a kernel that is bound by the vector instructions it issues: measured 4 % slower.
    float2 g0[AFF ? KS : 1], g1[AFF ? KS : 1];
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(AFF ? p.ab_in + (size_t)b * p.cin * 2 : p.wt), 0, AFF ? p.cin * 8 : 0, 0x00020000);
    const int gvoff = (lane >> 5) * 8;
    const int nsets = (p.cin + 2 * KS - 1) / (2 * KS);
#define PW_LOAD_SET(A, Bv, G, si)                                                                                        \
    _Pragma("unroll") for (int jq = 0; jq < KS / 4; ++jq)                                                               \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                                             \
            const float4 w4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wvoff[tm], ((si) * (KS / 4) + jq) * 1024, 0)); \
            A[tm][4 * jq + 0] = w4.x; A[tm][4 * jq + 1] = w4.y; A[tm][4 * jq + 2] = w4.z; A[tm][4 * jq + 3] = w4.w;       \
        }                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                                                    \
        if (PAIR) {                                                                                                      \
            const float2 x2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(xsrc, xvoff[0] + ((si) * KS + j) * xstep, 0, 0)); \
            Bv[0][j] = x2.x; Bv[TN - 1][j] = x2.y;                                                                       \
        } else {                                                                                                         \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) Bv[tn][j] = __builtin_bit_cast(                            \
                float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, SRC2 ? src2_off((si) * KS + j, tn) : xvoff[tn] + ((si) * KS + j) * xstep, 0, 0)); \
        }                                                                                                                \
        if (AFF) G[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(gsrc, gvoff + ((si) * KS + j) * 16, 0, 0)); \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
#define PW_MFMA_SET(A, Bv, G)                                                                                            \
    if (AFF) {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                                                \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                         \
                const float t = __builtin_fmaf(G[j].x, Bv[tn][j], G[j].y);                                               \
                Bv[tn][j] = relu_bits(t);                                                                                \
            }                                                                                                            \
        }                                                                                                                \
    }                                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < KS; ++j)                                                                      \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                               \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                           \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[tm][j], Bv[tn][j], acc[tm][tn], 0, 0, 0);           \
    __builtin_amdgcn_sched_barrier(0);
    PW_LOAD_SET(a0, b0, g0, 0)
    for (int c = 0; c + 1 < nsets; c += 2) {
        PW_LOAD_SET(a1, b1, g1, c + 1)
        PW_MFMA_SET(a0, b0, g0)
        PW_LOAD_SET(a0, b0, g0, (c + 2 < nsets ? c + 2 : nsets - 1))
        PW_MFMA_SET(a1, b1, g1)
    }
    if (nsets & 1) { PW_MFMA_SET(a0, b0, g0) }
#undef PW_LOAD_SET
#undef PW_MFMA_SET

    if constexpr (ST && PAIR) {
        // the rotation heads' layers (whole tiles, channel-major output, no activation on this path): the epilogue with nothing to
        // decide per element.  Everything a wave issues here is time its SIMD's matrix pipe does not get back (ablations, DESIGN
        // 3.2): 32 stores + ~280 vector instructions per 64 x 64 tile instead of 96 stores + ~1000
        if (!p.y_pm && pos0 + 64 <= p.L && co0 + TM * 32 <= p.cout && dbg == 0) {
            const int h = lane >> 5;
            const int tcol = (int)(pos0 / 64);
            const int rr = ((lane >> 1) & 1) + 2 * (lane & 1) + 4 * ((lane >> 3) & 1) + 8 * ((lane >> 2) & 1);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row0 = co0 + tm * 32 + 4 * h;
                float *yp = p.y + ((size_t)b * p.cout + row0) * p.L + pos0 + 2 * (lane & 31);
                float vs[16], vq[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v0 = acc[tm][0][r], v1 = acc[tm][1][r];
                    *reinterpret_cast<float2 *>(yp + (size_t)((r & 3) + 8 * (r >> 2)) * p.L) = make_float2(v0, v1);
                    vs[r] = v0 + v1;
                    vq[r] = v0 * v0 + v1 * v1;
                }
                const float sm = half_wave_transpose_sum(vs, lane), sq = half_wave_transpose_sum(vq, lane);
                if ((lane & 16) == 0)
                    *reinterpret_cast<float2 *>(p.stats_out + (((size_t)b * p.cout + row0 + (rr & 3) + 8 * (rr >> 2)) * p.stats_t + tcol) * 2) =
                        make_float2(sm, sq);
            }
            return;
        }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row0 = co0 + tm * 32 + 4 * (lane >> 5);
        if ((dbg & 1) && acc[tm][0][0] != 1.2345e-30f) continue;
        if (PAIR && !p.y_pm) {
            // both column tiles' outputs of a row are neighbours: one 8-byte store (col even, L even: aligned)
            const long long col = col_of(0);
            if (col < p.L) {
                float *yp = p.y + ((size_t)b * p.cout + row0) * p.L + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (row0 + ro < p.cout)
                        *reinterpret_cast<float2 *>(yp + (size_t)ro * p.L) =
                            make_float2(apply_act(acc[tm][0][r], p.act), apply_act(acc[tm][TN - 1][r], p.act));
                }
            }
            continue;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const long long col = col_of(tn);
            if (col < p.L && p.y_pm) {
                // point-major output: registers 4q..4q+3 are four consecutive channels of this lane's position
                float *yp = p.y + ((size_t)b * p.L + col) * p.cout + row0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (row0 + 8 * q + 3 < p.cout)
                        *reinterpret_cast<float4 *>(yp + 8 * q) =
                            make_float4(apply_act(acc[tm][tn][4 * q + 0], p.act), apply_act(acc[tm][tn][4 * q + 1], p.act),
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
    if (ST && !(dbg & 2)) {
        float dsm = 0.f, dsq = 0.f;   // (dbg & 8: one statistics store per row tile, timing only)
        // (sum, sum of squares) of the RAW outputs (act is none on this path) of every row over this wave's TN*32 columns
        const int tcol = (int)(pos0 / (TN * 32));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row0 = co0 + tm * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float v = (col_of(tn) < p.L) ? acc[tm][tn][r] : 0.f;
                    sm += v;
                    sq += v * v;
                }
                sm = half_wave_sum(sm);
                sq = half_wave_sum(sq);
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                if (dbg & 8) { dsm += sm; dsq += sq; if (r < 15) continue; sm = dsm; sq = dsq; }
                if ((lane & 31) == 16 && row < p.cout) {
                    float *d = p.stats_out + (((size_t)b * p.cout + row) * p.stats_t + tcol) * 2;
                    d[0] = sm;
                    d[1] = sq;
                }
            }
        }
    }
}

// SPLIT-K form of the direct kernel for launches with FEW positions (one or two trajectories: the 128- / 512-point levels are 4-16
// position tiles per cloud, and a 32x32 wave tile is ONE dependent MFMA chain of cin / 2 steps -- 768 of them, 23 us of matrix
// pipe, for FP3's first layer -- on a chip that is otherwise idle).  The four waves of a workgroup own the SAME 32 x 32 output tile
// and a quarter of the k-steps each (whole quads of the fragment image); partial tiles meet in LDS and are added in wave order,
// ((p0 + p1) + p2) + p3 with the bias in p0: a fixed order, but not the single k-ascending chain -- results differ from the
// bit-exact form in the last bits (tested at 1e-5 relative against it), so the launchers take this form only where the caller
// asked for it (captra_pw_set_splitk: EvalTrackModel at <= 2 trajectories; north_star's tolerance is 1e-4 on the poses).
// AFF / ST: the layer inside a Conv -> GroupNorm -> ReLU chain (captra_pointwise_mlp_gn): the operand is relu(a x + b) per (cloud,
// input channel), the epilogue also writes the raw output's (sum, sum of squares) per row and 32-column tile.
template <bool SRC2, bool AFF = false, bool ST = false>
__global__ __launch_bounds__(256) void pw_splitk_kernel(PwParams p) {
    static_assert(!SRC2 || (!AFF && !ST), "two-source input: plain layers only");
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, co0 = blockIdx.y * 32;
    const long long pos0 = (long long)blockIdx.x * 32;
    const int kp = (p.cin + 31) / 32 * 32;
    const int kq = ((p.cin + 1) / 2 + 3) / 4;
    const int ntile = (p.cout + 31) / 32;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.wt + (size_t)kp * p.ldw), 0, ntile * kq * 1024, 0x00020000);
    const float *xb = p.x + (size_t)b * (SRC2 ? p.csplit : p.cin) * p.L;
    const float *x2b = SRC2 ? p.x2 + (size_t)b * (p.cin - p.csplit) * (p.x2_bcast ? 1 : p.L) : nullptr;
    const float *xbase = SRC2 && x2b < xb ? x2b : xb;
    const int off1 = SRC2 ? (int)((const char *)xb - (const char *)xbase) : 0, off2 = SRC2 ? (int)((const char *)x2b - (const char *)xbase) : 0;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xbase, 0, SRC2 ? 0x7ffffff0 : (int)((long long)p.cin * p.L * 4), 0x00020000);
    long long col = pos0 + (lane & 31);
    const bool col_ok = col < p.L;
    if (!col_ok) col = p.L - 1;                                   // clamped column: computed, never stored
    const int col4 = (int)(col * 4), rowb = (int)(p.L * 4);
    auto xoff = [&](int ks) {                                     // byte offset of operand row 2 ks + h, this lane's column
        const int r = 2 * ks + h;
        if (!SRC2) return r * rowb + col4;                        // rows >= cin lie beyond the descriptor: read as 0
        if (r >= p.cin) return 0x7ffffffc;
        return r < p.csplit ? off1 + r * rowb + col4 : off2 + (r - p.csplit) * (p.x2_bcast ? 4 : rowb) + (p.x2_bcast ? 0 : col4);
    };
    const int q0 = (int)((long long)wave * kq / 4), q1 = (int)((long long)(wave + 1) * kq / 4);
    const int woff = blockIdx.y * kq * 1024 + lane * 16;
    f32x16 acc;
    {
        const float *bp = p.bias + (size_t)b * p.bias_bs + co0 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = wave == 0 ? bp[(r & 3) + 8 * (r >> 2)] : 0.f;
    }
    float4 a[2];
    float bv[2][4];
    float2 gv[2][AFF ? 4 : 1];
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(AFF ? p.ab_in + (size_t)b * p.cin * 2 : p.wt), 0, AFF ? p.cin * 8 : 0, 0x00020000);
#define SK_LOAD(s, q)                                                                                                    \
    a[s] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, woff, (q) * 1024, 0));                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
        bv[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, xoff(4 * (q) + j), 0, 0));       \
        if (AFF) gv[s][j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(gsrc, h * 8 + (4 * (q) + j) * 16, 0, 0)); \
    }
#define SK_MFMA(s)                                                                                                       \
    if (AFF) {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bv[s][j] = relu_bits(__builtin_fmaf(gv[s][j].x, bv[s][j], gv[s][j].y)); \
    }                                                                                                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].x, bv[s][0], acc, 0, 0, 0);                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].y, bv[s][1], acc, 0, 0, 0);                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].z, bv[s][2], acc, 0, 0, 0);                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].w, bv[s][3], acc, 0, 0, 0);
    if (q0 < q1) {
        SK_LOAD(0, q0)
        int q = q0;
        for (; q + 1 < q1; q += 2) {
            SK_LOAD(1, q + 1)
            SK_MFMA(0)
            SK_LOAD(0, (q + 2 < q1 ? q + 2 : q1 - 1))
            SK_MFMA(1)
        }
        if (q < q1) { SK_MFMA(0) }
    }
#undef SK_LOAD
#undef SK_MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave w finishes accumulator registers 4 w .. 4 w + 3 = rows co0 + 8 w + 4 h + {0, 1, 2, 3} of this lane's column
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i;
        v[i] = apply_act(((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane], p.act);
    }
    const int row0 = co0 + 8 * wave + 4 * h;
    if (ST) {
        // (sum, sum of squares) of the raw outputs (act is none on this path) of rows row0 .. row0 + 3 over the tile's 32 columns
        const int tcol = (int)(pos0 / 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = col_ok ? v[i] : 0.f;
            const float sm = half_wave_sum(t), sq = half_wave_sum(t * t);
            if ((lane & 31) == 16 && row0 + i < p.cout) {
                float *d = p.stats_out + (((size_t)b * p.cout + row0 + i) * p.stats_t + tcol) * 2;
                d[0] = sm;
                d[1] = sq;
            }
        }
    }
    if (!col_ok) return;
    if (p.y_pm) {
        float *yp = p.y + ((size_t)b * p.L + col) * p.cout + row0;
        if (row0 + 3 < p.cout) *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int i = 0; i < 4; ++i)
                if (row0 + i < p.cout) yp[i] = v[i];
        return;
    }
    float *yp = p.y + ((size_t)b * p.cout + row0) * p.L + col;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (row0 + i < p.cout) yp[(size_t)i * p.L] = v[i];
}

// Dense layer + ReLU + max over groups of K consecutive positions (K in {32, 64, 128}; SA3's last layer over its 128 points,
// reference pointnet_utils.py:336-343), direct-operand form: a workgroup = 4 waves x 32 positions x one 32-row output
// tile, both operands prefetched 16 k-steps ahead (one accumulator chain per wave, so the sets are deep), ReLU and the
// 32-lane max on the bit patterns (v_max_i32_dpp), the 4 waves' maxima combined through LDS.
__global__ __launch_bounds__(256) void pw_direct_max_kernel(PwParams p) {
    constexpr int KS = 16;
    __shared__ float red[32 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * 32;
    const long long blk0 = (long long)blockIdx.x * 128;
    const long long pos0 = blk0 + wave * 32;
    const int kp = (p.cin + 31) / 32 * 32;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.wt, 0, kp * p.ldw * 4, 0x00020000);
    const float *xb = p.x + (size_t)b * p.cin * p.L;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.cin * p.L * 4), 0x00020000);
    const int wstep = 2 * p.ldw * 4, xstep = (int)(2 * p.L * 4);
    const int wvoff = (((lane >> 5) * p.ldw) + co0 + (lane & 31)) * 4;
    long long col = pos0 + (lane & 31);
    const bool col_ok = col < p.L;
    if (!col_ok) col = p.L - 1;
    const int xvoff = (int)(((long long)(lane >> 5) * p.L + col) * 4);
    f32x16 acc;
    {
        const float *bp = p.bias + co0 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bp[(r & 3) + 8 * (r >> 2)];
    }
    float a0[KS], a1[KS], b0[KS], b1[KS];
    const int nsets = (p.cin + 2 * KS - 1) / (2 * KS);
#define PM_LOAD(A, Bv, si)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                                                    \
        A[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wsrc, wvoff, ((si) * KS + j) * wstep, 0)); \
        Bv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, xvoff + ((si) * KS + j) * xstep, 0, 0)); \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
#define PM_MFMA(A, Bv)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < KS; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j], Bv[j], acc, 0, 0, 0);  \
    __builtin_amdgcn_sched_barrier(0);
    PM_LOAD(a0, b0, 0)
    for (int c = 0; c + 1 < nsets; c += 2) {
        PM_LOAD(a1, b1, c + 1)
        PM_MFMA(a0, b0)
        PM_LOAD(a0, b0, (c + 2 < nsets ? c + 2 : nsets - 1))
        PM_MFMA(a1, b1)
    }
    if (nsets & 1) { PM_MFMA(a0, b0) }
#undef PM_LOAD
#undef PM_MFMA
    int v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int xbits = __float_as_int(acc[r]);
        v[r] = (xbits > 0 && col_ok) ? xbits : 0;       // ReLU on the bit pattern; columns beyond L contribute 0
    }
#define PM_STEP(CTRL)                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
        const int o = __builtin_amdgcn_update_dpp(0, v[r], CTRL, 0xF, 0xF, true);                       \
        v[r] = o > v[r] ? o : v[r];                                                                     \
    }
    PM_STEP(0xB1) PM_STEP(0x4E) PM_STEP(0x141) PM_STEP(0x140)
#undef PM_STEP
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = __builtin_amdgcn_update_dpp(v[r], v[r], 0x142, 0xA, 0xF, false);  // row_bcast15 into rows 1, 3
        v[r] = o > v[r] ? o : v[r];
    }
    if ((lane & 31) == 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 4 + wave] = __int_as_float(v[r]);
    }
    __syncthreads();
    const int tiles_per_group = p.k / 32, groups = 128 / p.k;
    for (int e = tid; e < 32 * groups; e += 256) {
        const int row = e / groups, gi = e % groups;
        const long long g = blk0 / p.k + gi;
        if (g < p.m && co0 + row < p.cout) {
            float mx = red[row * 4 + gi * tiles_per_group];
            for (int t = 1; t < tiles_per_group; ++t) mx = fmaxf(mx, red[row * 4 + gi * tiles_per_group + t]);
            p.y[((size_t)b * p.y_ctotal + p.co_off + co0 + row) * p.m + g] = mx;
        }
    }
}

// One wave per (cloud, group): fixed-order double-precision sum of the partials -> mean, rstd -> per-channel (a, b) with
// GroupNorm(x) = a*x + b  (a = gamma*rstd, b = beta - mean*a).
// TILE_MAJOR: stats (B,t,c,2) instead of (B,c,t,2) (the bf16 dense layers' statistics epilogue writes a wave's 32 channels of
// one chunk as 256 contiguous bytes).
template <bool TILE_MAJOR>
__global__ __launch_bounds__(256) void gn_finalize_kernel(int nb, int c, int cpg, int t, long long n, float eps,
                                                          const float *__restrict__ stats, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ ab) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int groups = c / cpg;
    if (e >= nb * groups) return;  // wave-uniform
    const int bi = e / groups, g = e % groups;
    double sm = 0.0, sq = 0.0;
    if (TILE_MAJOR) {
        const float2 *s2 = reinterpret_cast<const float2 *>(stats + (size_t)bi * c * t * 2) + (size_t)g * cpg;
        for (int i = lane; i < cpg * t; i += 64) {
            const float2 v = s2[(size_t)(i / cpg) * c + (i % cpg)];
            sm += (double)v.x;
            sq += (double)v.y;
        }
    } else {
        // the group's partials are contiguous: cpg channels x t tiles x (sum, sumsq)
        const float2 *s2 = reinterpret_cast<const float2 *>(stats + ((size_t)bi * c + (size_t)g * cpg) * t * 2);
        for (int i = lane; i < cpg * t; i += 64) {
            const float2 v = s2[i];
            sm += (double)v.x;
            sq += (double)v.y;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sm += __shfl_xor(sm, off, 64);
        sq += __shfl_xor(sq, off, 64);
    }
    const double cnt = (double)cpg * (double)n;
    const double mean = sm / cnt;
    double var = sq / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int ch = g * cpg + lane; ch < (g + 1) * cpg; ch += 64) {
        const double a = (double)gamma[ch] * rstd;
        ab[((size_t)bi * c + ch) * 2 + 0] = (float)a;
        ab[((size_t)bi * c + ch) * 2 + 1] = (float)((double)beta[ch] - mean * a);
    }
}

static CAPTRA_KNOB int g_pw_direct = 1;  // experiment knob: 0 = LDS-staged kernel for dense layers too
// Occupancy of the 64x64-wave-tile launches.  The kernels are built for FOUR workgroups per CU (<= 128 registers,
// amdgpu_waves_per_eu(4); the first build took 148-160 and held three).  Every workgroup computes one equal tile and a launch
// ends with its last ROUND of workgroups: 4096 tiles over 3 x 256 slots were 5.33 rounds paid as 6, over 4 x 256 they are 4.
// tools/bench_dense.py (512 -> 512 at 32 clouds, 4 / 3 / 2 workgroups per CU): GroupNorm-consuming layers 577 / 593 / 671 us,
// plain layers 651 / 630 / 570 us on random operands — but in the track step (real operands, the other network alongside)
// two per CU loses everywhere (5320 against 5380 frames/s), so four it is; the knob below (a launch limits its own
// residency by asking for unused dynamic LDS) stays for such measurements.
static CAPTRA_KNOB int g_pw_occ = 0;       // experiment knob: 0 / 4 = as built, 3 / 2 = fewer workgroups per CU
extern "C" void captra_pw_set_occupancy(int occ) { g_pw_occ = occ; }
static inline unsigned pw_occupancy_pad() { return g_pw_occ == 2 ? 60000u : (g_pw_occ == 3 ? 45000u : 0u); }
// split-K form for launches of at most this many positions (b * l) and at least 128 input channels; 0 = never (default: every
// layer is the k-ascending chain).  Set by EvalTrackModel for steps of one or two trajectories.
// (captra_launch_opts::splitk_positions of the call -- no state in the library: include/captra_hip.h section 3)
static inline int pw_splitk_of(const captra_launch_opts *o) { return (o != nullptr && o->splitk_positions > 0) ? o->splitk_positions : 0; }
static inline bool pw_use_splitk(int splitk_pos, int b, const PwParams &p) {
    return splitk_pos > 0 && (long long)b * p.L <= splitk_pos && p.cin >= 128 && (p.y_pm == 0 || p.cout % 4 == 0) &&
           (long long)p.cin * p.L * 4 < (1ll << 31);
}
template <bool SRC2, bool AFF = false, bool ST = false>
static int launch_pw_splitk(int b, const PwParams &p, hipStream_t s) {
    // (ST: the statistics table has an even number of 32-column tiles per row -- the 64-position workgroups of the chain form wrote
    // both; a tile beyond the last position writes zeros there and stores nothing else)
    dim3 grid((unsigned)(ST ? (p.L + 63) / 64 * 2 : (p.L + 31) / 32), (p.cout + 31) / 32, b);
    CAPTRA_LAUNCH("pointwise_mlp", (pw_splitk_kernel<SRC2, AFF, ST>), grid, dim3(256), 0, s, p);
    return captra_last_error();
}
static CAPTRA_KNOB int g_pw_dbg = 0;     // CAPTRA_ABLATIONS builds: PwParams::dbg of captra_pointwise_mlp_gn's launches
extern "C" void captra_pw_set_dbg(int v) { g_pw_dbg = CAPTRA_ABLATIONS ? v : 0; }
static CAPTRA_KNOB int g_pw_pair = 1;    // experiment knob: 0 = never the paired-column variant
extern "C" void captra_pw_set_pair(int on) { g_pw_pair = on; }
// paired column tiles need 8-byte aligned row segments: even L, 8-byte aligned tensors
static inline bool pw_pairable(const PwParams &p) {
    return g_pw_pair && p.L % 2 == 0 && ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.y)) & 7) == 0;
}

int launch_pw_direct(int b, const PwParams &p, hipStream_t s) {
    if ((long long)p.cin * p.L * 4 >= (1ll << 31)) return -3;  // buffer offsets are 32-bit: fall back
    // 64x64 wave tiles (4 accumulator chains) unless that leaves most SIMDs without a wave: the small-L layers
    // (SA3 / FP3 / FP2: 128-512 points per cloud) then take 32x32 wave tiles, 4x the waves, same bits
    const long long waves22 = ((p.L + 63) / 64) * ((p.cout + 63) / 64) * b;
    if (p.cout > 64 && waves22 < 2048) {
        dim3 grid((unsigned)((p.L + 63) / 64), (p.cout + 63) / 64, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 1, 2, 2>), grid, dim3(256), 0, s, p);
    } else if (p.cout > 64) {
        dim3 grid((unsigned)((p.L + 127) / 128), (p.cout + 127) / 128, b);
        if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, false, false, true>), grid, dim3(256), pw_occupancy_pad(), s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2>), grid, dim3(256), pw_occupancy_pad(), s, p); }
    } else if (p.cout > 32) {
        dim3 grid((unsigned)((p.L + 255) / 256), 1, b);
        if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 1, 4, false, false, true>), grid, dim3(256), 0, s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 1, 4>), grid, dim3(256), 0, s, p); }
    } else {
        dim3 grid((unsigned)((p.L + 255) / 256), 1, b);
        if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 2, 1, 4, false, false, true>), grid, dim3(256), 0, s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 2, 1, 4>), grid, dim3(256), 0, s, p); }
    }
    return captra_last_error();
}

__global__ void pack_weights_kernel(int cin, int cout, int kp, int cp, const float *__restrict__ wt,
                                    const float *__restrict__ bias, float *__restrict__ wt_packed,
                                    float *__restrict__ bias_packed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < kp * cp) {
        const int k = e / cp, c = e % cp;
        wt_packed[e] = (k < cin && c < cout) ? wt[(size_t)k * cout + c] : 0.f;
    }
    if (e < cp) bias_packed[e] = e < cout ? bias[e] : 0.f;
}

// fragment order of a packed layer (include/captra_hip.h: captra_pack_weights_frag)
__global__ void pack_weights_frag_kernel(int cin, int ldw, int kq, int total, const float *__restrict__ wt_packed,
                                         float *__restrict__ wfrag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int i = e & 3, lane = (e >> 2) & 63, tq = e >> 8;
    const int q = tq % kq, t = tq / kq;
    const int row = 2 * (4 * q + i) + (lane >> 5), col = 32 * t + (lane & 31);
    wfrag[e] = (row < cin && col < ldw) ? wt_packed[(size_t)row * ldw + col] : 0.f;
}

}  // namespace

extern "C" void captra_pw_set_direct(int on) { g_pw_direct = on; }

extern "C" long long captra_packed_weight_floats(int cin, int cout) {
    if (cin < 1 || cout < 1) return 0;
    const long long kp = (cin + 31) / 32 * 32, cp = (cout + 127) / 128 * 128;
    const long long kst = (cin + 1) / 2, kq = (kst + 3) / 4, nt = (cout + 31) / 32;
    return kp * cp + nt * kq * 256;
}

extern "C" int captra_pack_weights_frag(int cin, int cout, float *wt_packed, captra_stream_t stream) {
    if (cin < 1 || cout < 1) return -1;
    const int kp = (cin + 31) / 32 * 32, ldw = (cout + 127) / 128 * 128;
    const int kst = (cin + 1) / 2, kq = (kst + 3) / 4, nt = (cout + 31) / 32;
    const int total = nt * kq * 256;
    CAPTRA_LAUNCH("pack_weights", pack_weights_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, cin, ldw,
                  kq, total, wt_packed, wt_packed + (size_t)kp * ldw);
    return captra_last_error();
}

extern "C" int captra_pack_weights(int cin, int cout, const float *wt, const float *bias, float *wt_packed,
                                   float *bias_packed, captra_stream_t stream) {
    if (cin < 1 || cout < 1) return -1;
    const int kp = (cin + 31) / 32 * 32, cp = (cout + 127) / 128 * 128;
    const int n = kp * cp;
    CAPTRA_LAUNCH("pack_weights", pack_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cin,
                  cout, kp, cp, wt, bias, wt_packed, bias_packed);
    const int err = captra_last_error();
    return err != 0 ? err : captra_pack_weights_frag(cin, cout, wt_packed, stream);   // the fragment-ordered image behind it
}

extern "C" int captra_pointwise_mlp_ex(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                    const float *bias_packed, int act, float *y, const captra_launch_opts *opts, captra_stream_t stream) {
    const int splitk_pos = pw_splitk_of(opts);
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2) return -1;
    if (b == 0 || l == 0) return 0;
    PwParams p = {};
    p.cin = cin; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = l; p.x = x; p.wt = wt_packed; p.bias = bias_packed;
    p.y = y; p.act = act;
    if (pw_use_splitk(splitk_pos, b, p)) return launch_pw_splitk<false>(b, p, (hipStream_t)stream);
    if (g_pw_direct) {
        const int err = launch_pw_direct(b, p, (hipStream_t)stream);
        if (err != -3) return err;
    }
    const bool vec = (l % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec) return launch_pw<PRO_PLAIN, EPI_STORE, true>(b, p, (hipStream_t)stream, "pointwise_mlp");
    return launch_pw<PRO_PLAIN, EPI_STORE, false>(b, p, (hipStream_t)stream, "pointwise_mlp");
}
extern "C" int captra_pointwise_mlp(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                    const float *bias_packed, int act, float *y, captra_stream_t stream) {
    return captra_pointwise_mlp_ex(b, cin, cout, l, x, wt_packed, bias_packed, act, y, nullptr, stream);
}

// The layer with a bias PER CLOUD: bias_bc (B, ceil128(cout)), zero beyond cout -- y[b] = act(W x[b] + bias_bc[b]).  What a layer on
// [x; repeat(v)] (one vector v per cloud: pointnet_utils.py:265-270) becomes once the caller has formed W2 v + b per cloud; NOT the
// k-ascending chain over the concat (the f32x6 mode's FP3: captra_amd/pointnet_utils.py), per launch the same contract as
// captra_pointwise_mlp.  -2: shape outside the direct kernels.
extern "C" int captra_pointwise_mlp_cb(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                       const float *bias_bc, int act, float *y, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2 || bias_bc == nullptr) return -1;
    if (b == 0 || l == 0) return 0;
    PwParams p = {};
    p.cin = cin; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = l; p.x = x; p.wt = wt_packed; p.bias = bias_bc; p.bias_bs = p.ldw;
    p.y = y; p.act = act;
    const int err = launch_pw_direct(b, p, (hipStream_t)stream);
    return err == -3 ? -2 : err;
}

// The layer on the channel concat [x; x2] WITHOUT building it (SA3's [xyz, feat], pointnet_utils.py:171-188; FP3's
// [points1, repeat(points2)], pointnet_utils.py:265-270): x (B,csplit,L), x2 (B,cin - csplit,L), or (B,cin - csplit) with x2_bcast
// (one vector per cloud, the same for every position).  The operand rows are read in the concat's order, so the k-ascending chain
// -- and every output bit -- is that of captra_pointwise_mlp on the concatenated tensor.  -2 outside the direct kernel's small-launch
// shape or when the two tensors lie more than 2^30 bytes apart (the caller concatenates).
extern "C" int captra_pointwise_mlp2_ex(int b, int cin, int csplit, int cout, long long l, const float *x, const float *x2, int x2_bcast,
                                     const float *wt_packed, const float *bias_packed, int act, float *y, const captra_launch_opts *opts, captra_stream_t stream) {
    const int splitk_pos = pw_splitk_of(opts);
    if (b < 0 || cin < 2 || csplit < 1 || csplit >= cin || cout < 1 || l < 0 || act < 0 || act > 2 || x2 == nullptr) return -1;
    if (b == 0 || l == 0) return 0;
    const long long span1 = (long long)b * csplit * l * 4, span2 = (long long)b * (cin - csplit) * (x2_bcast ? 1 : l) * 4;
    const long long gap = (const char *)x2 > (const char *)x ? (const char *)x2 - (const char *)x : (const char *)x - (const char *)x2;
    if (gap + span1 + span2 >= (1ll << 30)) return -2;
    const long long waves22 = ((l + 63) / 64) * ((cout + 63) / 64) * b;
    if (cout <= 64) return -2;
    PwParams p = {};
    p.cin = cin; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = l; p.x = x; p.wt = wt_packed; p.bias = bias_packed;
    p.y = y; p.act = act; p.x2 = x2; p.csplit = csplit; p.x2_bcast = x2_bcast;
    if (pw_use_splitk(splitk_pos, b, p)) return launch_pw_splitk<true>(b, p, (hipStream_t)stream);
    if (waves22 < 2048) {
        dim3 grid((unsigned)((l + 63) / 64), (cout + 63) / 64, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 1, 2, 2, false, false, false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        dim3 grid((unsigned)((l + 127) / 128), (cout + 127) / 128, b);
        CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, false, false, false, true>), grid, dim3(256), pw_occupancy_pad(), (hipStream_t)stream, p);
    }
    return captra_last_error();
}
extern "C" int captra_pointwise_mlp2(int b, int cin, int csplit, int cout, long long l, const float *x, const float *x2, int x2_bcast,
                                     const float *wt_packed, const float *bias_packed, int act, float *y, captra_stream_t stream) {
    return captra_pointwise_mlp2_ex(b, cin, csplit, cout, l, x, x2, x2_bcast, wt_packed, bias_packed, act, y, nullptr, stream);
}

// The same layer with a POINT-major result y (B,L,cout): what a consumer that gathers whole points reads with 16-byte loads
// (the SA2 scales' pre-transformed first layer, csrc/sa_pipe.hip).  Direct-operand kernel only: -2 outside its range.
extern "C" int captra_pointwise_mlp_pm_ex(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                       const float *bias_packed, int act, float *y, const captra_launch_opts *opts, captra_stream_t stream) {
    const int splitk_pos = pw_splitk_of(opts);
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2) return -1;
    if (cout % 4 != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0) return -2;
    if (b == 0 || l == 0) return 0;
    PwParams p = {};
    p.cin = cin; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = l; p.x = x; p.wt = wt_packed; p.bias = bias_packed;
    p.y = y; p.act = act; p.y_pm = 1;
    if (pw_use_splitk(splitk_pos, b, p)) return launch_pw_splitk<false>(b, p, (hipStream_t)stream);
    const int err = launch_pw_direct(b, p, (hipStream_t)stream);
    return err == -3 ? -2 : err;
}
extern "C" int captra_pointwise_mlp_pm(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                       const float *bias_packed, int act, float *y, captra_stream_t stream) {
    return captra_pointwise_mlp_pm_ex(b, cin, cout, l, x, wt_packed, bias_packed, act, y, nullptr, stream);
}

// Dense layer inside a Conv -> GroupNorm -> ReLU chain (see include/captra_hip.h).
extern "C" int captra_pointwise_mlp_gn_ex(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                       const float *bias_packed, const float *ab_in, int act, float *y, float *stats_out,
                                       int stats_t, const captra_launch_opts *opts, captra_stream_t stream) {
    const int splitk_pos = pw_splitk_of(opts);
    if (b < 0 || cin < 1 || cout < 1 || l < 0 || act < 0 || act > 2) return -1;
    if (stats_out != nullptr && act != ACT_NONE) return -1;           // statistics are those of the raw output
    if ((long long)cin * l * 4 >= (1ll << 31)) return -2;
    if (b == 0 || l == 0) return 0;
    PwParams p = {};
    p.cin = cin; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = l; p.x = x; p.wt = wt_packed; p.bias = bias_packed;
    p.y = y; p.act = act; p.ab_in = ab_in; p.stats_out = stats_out; p.stats_t = stats_t; p.dbg = g_pw_dbg;
    hipStream_t s = (hipStream_t)stream;
    const long long waves22 = ((l + 63) / 64) * ((cout + 63) / 64) * b;
    // (the same rule as captra_pointwise_mlp_gn_tiles: under captra_pw_set_splitk every launch within its position limit takes the
    // 32-column statistics tiles, split-k or -- fewer than 128 input channels -- the 32x32 chain form)
    if (cout > 64 && (waves22 < 2048 || (splitk_pos > 0 && (long long)b * l <= splitk_pos))) {
        // few positions (single-trajectory latency): 32x32 wave tiles, statistics per 32-column tile
        if (stats_out != nullptr && stats_t != (int)((l + 63) / 64) * 2) return -1;
        if (pw_use_splitk(splitk_pos, b, p)) {
            if (ab_in != nullptr && stats_out != nullptr) return launch_pw_splitk<false, true, true>(b, p, s);
            if (stats_out != nullptr) return launch_pw_splitk<false, false, true>(b, p, s);
            if (ab_in != nullptr) return launch_pw_splitk<false, true, false>(b, p, s);
            return launch_pw_splitk<false>(b, p, s);
        }
        dim3 grid((unsigned)((l + 63) / 64), (cout + 63) / 64, b);
        if (ab_in != nullptr && stats_out != nullptr) {
            CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 1, 2, 2, true, true>), grid, dim3(256), 0, s, p);
        } else if (stats_out != nullptr) {
            CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 1, 2, 2, false, true>), grid, dim3(256), 0, s, p);
        } else if (ab_in != nullptr) {
            CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 1, 2, 2, true, false>), grid, dim3(256), 0, s, p);
        } else {
            CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 1, 2, 2>), grid, dim3(256), 0, s, p);
        }
        return captra_last_error();
    }
    if (cout > 64) {
        if (stats_out != nullptr && stats_t != (int)((l + 127) / 128) * 2) return -1;
        dim3 grid((unsigned)((l + 127) / 128), (cout + 127) / 128, b);
        if (ab_in != nullptr && stats_out != nullptr) {
            if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, true, true, true>), grid, dim3(256), pw_occupancy_pad(), s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, true, true>), grid, dim3(256), pw_occupancy_pad(), s, p); }
        } else if (stats_out != nullptr) {
            if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, false, true, true>), grid, dim3(256), pw_occupancy_pad(), s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, false, true>), grid, dim3(256), pw_occupancy_pad(), s, p); }
        } else if (ab_in != nullptr) {
            if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, true, false, true>), grid, dim3(256), pw_occupancy_pad(), s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, true, false>), grid, dim3(256), pw_occupancy_pad(), s, p); }
        } else {
            if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2, false, false, true>), grid, dim3(256), pw_occupancy_pad(), s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 2, 2>), grid, dim3(256), pw_occupancy_pad(), s, p); }
        }
        return captra_last_error();
    }
    if (stats_out != nullptr) return -2;   // statistics only from the 64x64 wave-tile configuration
    // few positions and few output channels (a head's last layer at one trajectory: 16 workgroups, each wave a chain of cin / 2 steps)
    if (pw_use_splitk(splitk_pos, b, p)) return ab_in != nullptr ? launch_pw_splitk<false, true, false>(b, p, s) : launch_pw_splitk<false>(b, p, s);
    dim3 grid((unsigned)((l + 255) / 256), 1, b);
    if (cout > 32) {
        if (ab_in != nullptr) {
            if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 1, 4, true, false, true>), grid, dim3(256), 0, s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 1, 4, true, false>), grid, dim3(256), 0, s, p); }
        } else {
            if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 1, 4, false, false, true>), grid, dim3(256), 0, s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<2, 2, 1, 4>), grid, dim3(256), 0, s, p); }
        }
    } else if (ab_in != nullptr) {
        if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 2, 1, 4, true, false, true>), grid, dim3(256), 0, s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 2, 1, 4, true, false>), grid, dim3(256), 0, s, p); }
    } else {
        if (pw_pairable(p)) { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 2, 1, 4, false, false, true>), grid, dim3(256), 0, s, p); } else { CAPTRA_LAUNCH("pointwise_mlp", (pw_direct_kernel<1, 2, 1, 4>), grid, dim3(256), 0, s, p); }
    }
    return captra_last_error();
}
extern "C" int captra_pointwise_mlp_gn(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                                       const float *bias_packed, const float *ab_in, int act, float *y, float *stats_out,
                                       int stats_t, captra_stream_t stream) {
    return captra_pointwise_mlp_gn_ex(b, cin, cout, l, x, wt_packed, bias_packed, ab_in, act, y, stats_out, stats_t, nullptr, stream);
}

extern "C" int captra_pointwise_mlp_gn_tiles_ex(int b, int cout, long long l, const captra_launch_opts *opts) {
    const int splitk_pos = pw_splitk_of(opts);
    const long long waves22 = ((l + 63) / 64) * ((cout + 63) / 64) * b;
    const bool splitk = splitk_pos > 0 && (long long)b * l <= splitk_pos;       // (the split-k form: 32-column tiles; cin >= 128 is the caller's layer)
    return (cout > 64 && (waves22 < 2048 || splitk)) ? (int)((l + 63) / 64) * 2 : (int)((l + 127) / 128) * 2;
}
extern "C" int captra_pointwise_mlp_gn_tiles(int b, int cout, long long l) { return captra_pointwise_mlp_gn_tiles_ex(b, cout, l, nullptr); }

extern "C" int captra_gn_finalize(int b, int c, int channels_per_group, int stats_t, long long n, float eps,
                                  const float *stats, const float *gamma, const float *beta, float *ab,
                                  captra_stream_t stream) {
    if (b < 0 || c < 1 || channels_per_group < 1 || c % channels_per_group != 0 || stats_t < 1 || n < 1) return -1;
    if (b == 0) return 0;
    const int total = b * (c / channels_per_group);
    CAPTRA_LAUNCH("gn_finalize", gn_finalize_kernel<false>, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, b, c,
                  channels_per_group, stats_t, n, eps, stats, gamma, beta, ab);
    return captra_last_error();
}

// The same for TILE-major partials, stats (B,stats_t,c,2) (captra_pointwise_mlp_bf16pm_stats).
extern "C" int captra_gn_finalize_tm(int b, int c, int channels_per_group, int stats_t, long long n, float eps,
                                     const float *stats, const float *gamma, const float *beta, float *ab,
                                     captra_stream_t stream) {
    if (b < 0 || c < 1 || channels_per_group < 1 || c % channels_per_group != 0 || stats_t < 1 || n < 1) return -1;
    if (b == 0) return 0;
    const int total = b * (c / channels_per_group);
    CAPTRA_LAUNCH("gn_finalize", gn_finalize_kernel<true>, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, b, c,
                  channels_per_group, stats_t, n, eps, stats, gamma, beta, ab);
    return captra_last_error();
}

extern "C" int captra_sa_group_mlp(int b, int n, int m, int k, int cfeat, int cout, const float *feat,
                                   const float *xyz_cn, const float *new_xyz, const int *idx, const float *wt_packed,
                                   const float *bias_packed, float *y, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 0 || cout < 1) return -1;
    if (cfeat > 0 && feat == nullptr) return -1;
    if (b == 0 || m == 0) return 0;
    PwParams p = {};
    p.cin = cfeat + 3; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = (long long)m * k; p.wt = wt_packed;
    p.bias = bias_packed; p.y = y; p.act = ACT_RELU;
    p.n = n; p.m = m; p.k = k; p.cfeat = cfeat; p.feat = feat; p.xyz_cn = xyz_cn; p.new_xyz = new_xyz; p.idx = idx;
    return launch_pw<PRO_GROUP, EPI_STORE, false>(b, p, (hipStream_t)stream, "sa_group_mlp");
}

extern "C" int captra_mlp_max(int b, int cin, int cout, int m, int k, const float *x, const float *wt_packed,
                              const float *bias_packed, float *y, int y_ctotal, int co_off, captra_stream_t stream) {
    if (b < 0 || cin < 1 || cout < 1 || m < 0 || k < 1 || y_ctotal < co_off + cout || co_off < 0) return -1;
    if (k % 32 != 0 || 128 % k != 0) return -2;  // fused max needs K in {32, 64, 128}
    if (b == 0 || m == 0) return 0;
    PwParams p = {};
    p.cin = cin; p.cout = cout; p.ldw = (cout + 127) / 128 * 128; p.L = (long long)m * k; p.x = x; p.wt = wt_packed;
    p.bias = bias_packed; p.y = y; p.act = ACT_RELU;
    p.m = m; p.k = k; p.y_ctotal = y_ctotal; p.co_off = co_off;
    if (g_pw_direct && (long long)cin * p.L * 4 < (1ll << 31)) {
        dim3 grid((unsigned)((p.L + 127) / 128), (cout + 31) / 32, b);
        CAPTRA_LAUNCH("mlp_max", pw_direct_max_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
        return captra_last_error();
    }
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0);  // L = m*k is a multiple of 32
    if (vec) return launch_pw<PRO_PLAIN, EPI_MAXK, true>(b, p, (hipStream_t)stream, "mlp_max");
    return launch_pw<PRO_PLAIN, EPI_MAXK, false>(b, p, (hipStream_t)stream, "mlp_max");
}
