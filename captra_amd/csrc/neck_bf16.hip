// The backbone's NECK in the bf16 mode -- SA3 (group_all: [xyz, feat] -> 256 -> 512 -> 1024, max over the 128 points), FP3 (the pooled
// vector as a per-cloud bias, 512 -> 256 -> 256 on 128 points) and FP2 (3-NN interpolation + skip concat, 576 -> 256 -> 128 on 512
// points): reference network/models/pointnet_utils.py:302-343, 253-299, backbones.py:52-66 -- as THREE launches instead of nine.
//
// Round 4 ran every layer as a launch of tb_layer_kernel (csrc/tile_bf16.hip) plus a gemv, two interpolation / concat kernels
// and point-major bf16 tensors in between: 0.15 ms per network for 2 % of the step's flops -- each launch a few exposed staging
// latencies on a mostly idle chip.  neck_chain_kernel runs a whole MODULE's layers on a tile of 64 positions: the first layer's
// input is staged through LDS in K-chunks of 128 channels exactly as tb_layer_kernel stages it (fp32 channel-major sources, the
// concat never built; for FP2 the interpolated channels are formed while they are staged; for FP3 the per-cloud bias W2 v + b is
// computed first, in tb_gemv_kernel's own summation order), every hidden activation lives in LDS as the next layer's B-operand
// image (bf16, slot order, [position][16-byte slots] with slot ^= position & 15: what a point-major tensor row is), weights stream
// from the layers' fragment images through a ring of four k-steps, and the last layer's epilogue stores fp32 channel-major rows
// or folds the tile's maximum into the pooled vector (integer atomic max on post-ReLU values: exact and order-free).
// Per layer the MFMA sequence, the bf16 roundings and the activation are tb_layer_kernel's: outputs are bit-identical to the
// layer-by-layer route (tests/test_neck_gpu.py).
#include "common.h"
#include "bf16_dense.h"

namespace {

constexpr int NK_P = 64;                 // positions per workgroup
constexpr int NK_NW = 8;                 // waves
constexpr int NK_NT = NK_NW * 64;
constexpr int NK_CHUNK = 16384;          // one staged K-chunk: 64 positions x 128 channels bf16
constexpr int NK_IMG_A = 2 * NK_CHUNK;   // first hidden image (<= 256 channels: 32 KiB)
constexpr int NK_IMG_B = NK_IMG_A + 32768;   // second hidden image (<= 512 channels: 64 KiB)
constexpr int NK_CB = NK_IMG_B + 65536;  // per-cloud bias (<= 256 floats)
constexpr int NK_LDS = NK_CB + 1024;

struct NkLayer {
    const unsigned char *wimg;           // captra_pack_dense_bf16(perm = 1): nt x kst fragments of 1 KiB
    const float *bias;                   // packed fp32 bias
    int cin, cout, kst, nt;
};
struct NkParams {
    long long L;                         // positions per cloud
    NkLayer ly[3];
    const float *x, *x2;                 // layer 1's input channels [0, csplit) from x (B,csplit,L), the rest from x2
    int csplit;
    // PRO 2 (FP2): x2 = known features (B,cin - csplit,S) interpolated through (idx, weight) (B,L,3) of captra_three_nn_weights
    const int *nn_idx;
    const float *nn_w;
    int s_known;
    // PRO 1 (FP3): layer 1's bias of cloud b = bias + W_v bf16(v[b]), v (B,cv), gw = W'^T rows of v's channels as bf16 (cv, gldw)
    // row-major (the RNE rounding of the packed fp32 rows: what captra_gemv_bf16 forms on the fly, at half the bytes per workgroup)
    const float *v;
    const unsigned short *gw;
    int cv, gldw;
    float *y;                            // EPI 0: (B,cout,L) fp32; EPI 1: (B,cout) fp32, zeroed by the launcher
    int act_last;
};

// MT row tiles x 2 column tiles per wave from an LDS image (row pitch `pitch` bytes, kst k-steps, kst % 8 == 0)
template <int MT>
__device__ __forceinline__ void nk_layer_from_image(f32x16 (&acc)[MT][2], const unsigned char *img, int pitch, const NkLayer &ly, const int (&woff)[MT],
                                                    int lane) {
    const int h = lane >> 5, col = lane & 31;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ly.wimg, 0, ly.nt * ly.kst * 1024, 0x00020000);
    const int kst = ly.kst;
    u32x4 A[4][MT];
    auto loadA = [&](int s, int kk) {
        kk = kk < kst ? kk : kst - 1;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, woff[tm] + kk * 1024, 0);
    };
    loadA(0, 0);
    loadA(1, 1);
    loadA(2, 2);
    const int e0 = (h ^ (col & 15)) << 4;
    const unsigned char *src = img + (size_t)col * pitch;
    u32x4 Bf[2][2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) Bf[0][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 32 * pitch + e0);
    for (int c = 0; c < (kst >> 3); ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * c + j;
            loadA((j + 3) & 3, kk + 3);
            {
                const int kn = kk + 1 < kst ? kk + 1 : kst - 1;        // (the last k-step re-reads itself: never multiplied)
                const int cn = kn >> 3, jn = (j + 1) & 7;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    Bf[(j + 1) & 1][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 32 * pitch + cn * 256 + ((32 * jn) ^ e0));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < MT; ++tm) acc[tm][tn] = db_mfma(A[j & 3][tm], Bf[j & 1][tn], acc[tm][tn]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// accumulators -> bf16(relu(.)) -> the image of the next layer (slot 4 t + 2 jj + h of position col)
template <int MT>
__device__ __forceinline__ void nk_park(const f32x16 (&acc)[MT][2], unsigned char *img, int pitch, int t0, int nt, int lane) {
    const int h = lane >> 5, col = lane & 31;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        const int t = t0 + tm;
        if (t >= nt) continue;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            unsigned char *row = img + (size_t)(tn * 32 + col) * pitch;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                u32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = db_relu2(db_pack(acc[tm][tn][8 * jj + 2 * i], acc[tm][tn][8 * jj + 2 * i + 1]));
                *reinterpret_cast<u32x4 *>(row + (((4 * t + 2 * jj + h) ^ (col & 15)) << 4)) = v;
            }
        }
    }
}

template <int MT>
__device__ __forceinline__ void nk_init_acc(f32x16 (&acc)[MT][2], const float *bias, int t0, int nt, int (&woff)[MT], int kst, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        const int t = t0 + tm < nt ? t0 + tm : nt - 1;                 // clamped row tile: computed, never parked / stored
        woff[tm] = t * kst * 1024;
        const float *bp = bias + t * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bp[(r & 3) + 8 * (r >> 2)];
            acc[tm][0][r] = bv;
            acc[tm][1][r] = bv;
        }
    }
}

// last layer's epilogue.  EPI 0: act + fp32 channel-major rows; EPI 1: the tile's maximum per channel into the pooled vector
template <int MT, int EPI>
__device__ __forceinline__ void nk_epilogue(const f32x16 (&acc)[MT][2], const NkParams &p, const NkLayer &ly, int b, long long pos0, int t0, int lane) {
    const int h = lane >> 5, col = lane & 31;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        const int t = t0 + tm;
        if (t >= ly.nt) continue;
        const int row0 = 32 * t + 4 * h;
        if constexpr (EPI == 1) {
            float mx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mx[r] = -INFINITY;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) mx[r] = pos0 + tn * 32 + col < p.L ? fmaxf(mx[r], acc[tm][tn][r]) : mx[r];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], off, 64));
            }
            if (col == 0) {
                int *yp = reinterpret_cast<int *>(p.y) + (size_t)b * ly.cout + row0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    // (post-ReLU values are non-negative floats: they order like their bit patterns; act(max) == max(act))
                    if (row0 + ro < ly.cout) atomicMax(yp + ro, __float_as_int(apply_act(mx[r], p.act_last)));
                }
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const long long c = pos0 + tn * 32 + col;
                if (c >= p.L) continue;
                float *yp = p.y + ((size_t)b * ly.cout + row0) * p.L + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (row0 + ro < ly.cout) yp[(size_t)ro * p.L] = apply_act(acc[tm][tn][r], p.act_last);
                }
            }
        }
    }
}

// PRO: 0 = plain two-source input, 1 = + per-cloud bias from the pooled vector (FP3), 2 = x2's channels interpolated (FP2)
// EPI: 0 = fp32 channel-major output, 1 = pooled maximum.  MT1..MT3: row tiles per wave of the layers (MT3 = 0: two layers)
// gridDim.z > 1: the LAST layer's row tiles are dealt to the z workgroups of a position tile (each recomputes the layers before it:
// a workgroup streams every weight it multiplies through ONE CU's L1 at what ~100 KB in flight buy, ~40 GB/s -- the last layer of SA3
// is 1 MB of the module's 1.57 MB)
template <int PRO, int EPI, int MT1, int MT2, int MT3>
__global__ __launch_bounds__(NK_NT, 1) void neck_chain_kernel(NkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;
    const int b = blockIdx.y;
    const long long pos0 = (long long)blockIdx.x * NK_P;
    const NkLayer &l1 = p.ly[0];
    float *cbias = reinterpret_cast<float *>(lds + NK_CB);

    if constexpr (PRO == 1) {
        // ---- the cloud's first-layer bias = bias + W_v bf16(v): tb_gemv_kernel's arithmetic and summation order (sixteen partial sums
        // over k = 16 i + w, added in order w = 0 .. 15), two of the sixteen per wave here ------------------------------------------
        float4 *red = reinterpret_cast<float4 *>(lds + NK_IMG_B);     // [16][64]
        const float *vb = p.v + (size_t)b * p.cv;
        const int co = lane * 4;
        const int cc = co < p.gldw ? co : p.gldw - 4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int vw = wave + 8 * r;
            float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
            for (int k = vw; k < p.cv; k += 16) {
                const float xv = (float)(__bf16)vb[k];
                const uint2 w = *reinterpret_cast<const uint2 *>(p.gw + (size_t)k * p.gldw + cc);
                acc.x = __builtin_fmaf(__uint_as_float(w.x << 16), xv, acc.x);
                acc.y = __builtin_fmaf(__uint_as_float(w.x & 0xffff0000u), xv, acc.y);
                acc.z = __builtin_fmaf(__uint_as_float(w.y << 16), xv, acc.z);
                acc.w = __builtin_fmaf(__uint_as_float(w.y & 0xffff0000u), xv, acc.w);
            }
            red[vw * 64 + lane] = acc;
        }
        __syncthreads();
        if (wave == 0) {
            float4 s = red[lane];
            for (int w = 1; w < 16; ++w) {
                const float4 t = red[w * 64 + lane];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            const float vv[4] = {s.x, s.y, s.z, s.w};
            for (int i = 0; i < 4; ++i)
                if (co + i < l1.cout) cbias[co + i] = l1.bias[co + i] + vv[i];
        }
        __syncthreads();
    }

    // ---- layer 1: input staged in K-chunks of 128 channels (16 KiB, double-buffered, one barrier per chunk) ------------------------
    // item = (position quad pq, slot sl) of a chunk: 16 x 16 = 256 items, one per thread of the lower half of the workgroup; a slot's
    // 8 channels are base + {0,1,2,3,8,9,10,11}, base = 16 kk + 4 hh (slot order)
    const int kst1 = l1.kst, nch = (kst1 + 7) >> 3;
    // (every thread stages HALF a slot -- 4 channels x 4 positions: e = 4 half .. 4 half + 3 -- so that the interpolated channels' 12
    // gathers per value are spread over all eight waves)
    const int pq = tid & 15, sl = (tid >> 4) & 15, half = tid >> 8;
    long long spos = pos0 + 4 * pq;
    if (spos > p.L - 4) spos = p.L - 4;                               // clamped quad: computed, never stored (L % 4 == 0, L >= 4)
    const float *x0b = p.x + (size_t)b * p.csplit * p.L;
    const int c2 = l1.cin - p.csplit;
    const float *x1b = PRO == 2 ? p.x2 + (size_t)b * c2 * p.s_known : p.x2 + (size_t)b * c2 * p.L;
    int nj[PRO == 2 ? 4 : 1][3];
    float nw[PRO == 2 ? 4 : 1][3];
    if constexpr (PRO == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int *ii = p.nn_idx + ((size_t)b * p.L + spos + r) * 3;
            const float *ww = p.nn_w + ((size_t)b * p.L + spos + r) * 3;
#pragma unroll
            for (int e = 0; e < 3; ++e) { nj[r][e] = ii[e]; nw[r][e] = ww[e]; }
        }
    }
    float4 sf[4];
    auto gload = [&](int c) {
        c = c < nch ? c : nch - 1;
        const int base = 16 * (8 * c + (sl >> 1)) + 4 * (sl & 1) + 8 * half;      // channels base + {0,1,2,3} (slot order: e = 4 half + i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int ch = base + e;
            ch = ch < l1.cin ? ch : l1.cin - 1;                       // clamped channel: loaded, zeroed when parked
            if (ch < p.csplit) {
                sf[e] = *reinterpret_cast<const float4 *>(x0b + (size_t)ch * p.L + spos);
            } else if constexpr (PRO == 2) {
                // interp_concat_kernel's expression: (w0 f[j0] + w1 f[j1]) + w2 f[j2], unfused
                const float *row = x1b + (size_t)(ch - p.csplit) * p.s_known;
                float q[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r] = (nw[r][0] * row[nj[r][0]] + nw[r][1] * row[nj[r][1]]) + nw[r][2] * row[nj[r][2]];
                sf[e] = make_float4(q[0], q[1], q[2], q[3]);
            } else {
                sf[e] = *reinterpret_cast<const float4 *>(x1b + (size_t)(ch - p.csplit) * p.L + spos);
            }
        }
    };
    auto park = [&](int c) {
        const int base = 16 * (8 * c + (sl >> 1)) + 4 * (sl & 1) + 8 * half;
        bool m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = base + e < l1.cin;
        unsigned char *dst = lds + (c & 1) * NK_CHUNK;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pos = 4 * pq + r;
            uint2 v;
            {
                const float a0 = r == 0 ? sf[0].x : r == 1 ? sf[0].y : r == 2 ? sf[0].z : sf[0].w;
                const float a1 = r == 0 ? sf[1].x : r == 1 ? sf[1].y : r == 2 ? sf[1].z : sf[1].w;
                const float a2 = r == 0 ? sf[2].x : r == 1 ? sf[2].y : r == 2 ? sf[2].z : sf[2].w;
                const float a3 = r == 0 ? sf[3].x : r == 1 ? sf[3].y : r == 2 ? sf[3].z : sf[3].w;
                v.x = db_pack(m[0] ? a0 : 0.f, m[1] ? a1 : 0.f);
                v.y = db_pack(m[2] ? a2 : 0.f, m[3] ? a3 : 0.f);
            }
            *reinterpret_cast<uint2 *>(dst + pos * 256 + ((sl ^ (pos & 15)) << 4) + 8 * half) = v;
        }
    };

    f32x16 acc1[MT1][2];
    int woff1[MT1];
    const int t01 = wave * MT1;
    nk_init_acc<MT1>(acc1, PRO == 1 ? cbias : l1.bias, t01, l1.nt, woff1, kst1, lane);
    {
        const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)l1.wimg, 0, l1.nt * kst1 * 1024, 0x00020000);
        u32x4 A[4][MT1];
        auto loadA = [&](int s, int kk) {
            kk = kk < kst1 ? kk : kst1 - 1;
#pragma unroll
            for (int tm = 0; tm < MT1; ++tm) A[s][tm] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, lane * 16, woff1[tm] + kk * 1024, 0);
        };
        gload(0);
        loadA(0, 0);
        loadA(1, 1);
        loadA(2, 2);
        park(0);
        gload(1);
        __syncthreads();
        const int e0 = (h ^ (col & 15)) << 4;
        const int brd = col * 256;
        u32x4 Bf[2][2];
        for (int c = 0; c < nch; ++c) {
            const unsigned char *src = lds + (c & 1) * NK_CHUNK + brd;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) Bf[0][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + e0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 8 * c + j;
                loadA((j + 3) & 3, kk + 3);
                if (j < 7 && kk + 1 < kst1) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) Bf[(j + 1) & 1][tn] = *reinterpret_cast<const u32x4 *>(src + tn * 8192 + ((32 * (j + 1)) ^ e0));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kk < kst1) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int tm = 0; tm < MT1; ++tm) acc1[tm][tn] = db_mfma(A[j & 3][tm], Bf[j & 1][tn], acc1[tm][tn]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j == 1 && c + 1 < nch) {                          // the next chunk: converted and parked under this chunk's MFMAs
                    park(c + 1);
                    gload(c + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
    }
    // ---- layer 1 -> image A -> layer 2 [-> image B -> layer 3] -> epilogue ----------------------------------------------------------
    const NkLayer &l2 = p.ly[1];
    const int pitch_a = l1.cout * 2;
    nk_park<MT1>(acc1, lds + NK_IMG_A, pitch_a, t01, l1.nt, lane);
    f32x16 acc2[MT2][2];
    int woff2[MT2];
    const int t02 = wave * MT2;
    nk_init_acc<MT2>(acc2, l2.bias, t02, l2.nt, woff2, l2.kst, lane);
    __syncthreads();
    nk_layer_from_image<MT2>(acc2, lds + NK_IMG_A, pitch_a, l2, woff2, lane);
    if constexpr (MT3 == 0) {
        nk_epilogue<MT2, EPI>(acc2, p, l2, b, pos0, t02, lane);        // (two-layer modules are launched with gridDim.z == 1)
    } else {
        const NkLayer &l3 = p.ly[2];
        const int pitch_b = l2.cout * 2;
        nk_park<MT2>(acc2, lds + NK_IMG_B, pitch_b, t02, l2.nt, lane);
        f32x16 acc3[MT3][2];
        int woff3[MT3];
        const int t03 = ((int)blockIdx.z * NK_NW + wave) * MT3;
        nk_init_acc<MT3>(acc3, l3.bias, t03, l3.nt, woff3, l3.kst, lane);
        __syncthreads();
        nk_layer_from_image<MT3>(acc3, lds + NK_IMG_B, pitch_b, l3, woff3, lane);
        nk_epilogue<MT3, EPI>(acc3, p, l3, b, pos0, t03, lane);
    }
}

template <int PRO, int EPI, int MT1, int MT2, int MT3>
int nk_launch(int b, const NkParams &p, hipStream_t s, int nz = 1) {
    auto kern = neck_chain_kernel<PRO, EPI, MT1, MT2, MT3>;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NK_LDS) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    CAPTRA_LAUNCH("neck_chain", kern, dim3((unsigned)((p.L + NK_P - 1) / NK_P), b, nz), dim3(NK_NT), NK_LDS, s, p);
    return captra_last_error();
}

CAPTRA_KNOB int g_nk_split = 4;          // measurement knob: workgroups sharing the last layer of a three-layer module (1 / 2 / 4)

}  // namespace

extern "C" void captra_neck_chain_set_split(int n) { g_nk_split = n; }

// A module of the backbone's neck (bf16 mode) in ONE launch.  nl = 2 or 3 layers, every layer act(b + W x) with ReLU between them; the
// layers' weights as captra_pack_dense_bf16(perm = 1) images wimg[i] + packed fp32 biases bias[i], channel counts c[0] (input) .. c[nl].
// Layer 1's input (B,c[0],L) is never built: channels [0, csplit) come from x (B,csplit,L) fp32 and
//   kind 0 (SA3, pointnet_utils.py:318-343): the rest from x2 (B,c[0] - csplit,L); y (B,c[nl]) = act_last(max over the L positions)
//          (y is zeroed here on `stream`; needs a non-negative act_last: CAPTRA_ACT_RELU);
//   kind 1 (FP3, pointnet_utils.py:265-298 with one source vector per cloud): csplit = c[0]; layer 1's bias of cloud b is bias[0] + W_v
//          bf16(v[b]) with v (B,cv) fp32 and gw (cv, ceil128(c[1])) bf16 = the RNE rounding of the packed fp32 W'^T rows of v's channels
//          -- the arithmetic of captra_gemv_bf16, which rounds the same rows on the fly; y (B,c[nl],L) fp32;
//   kind 2 (FP2, pointnet_utils.py:280-298): the rest interpolated from x2 (B,c[0] - csplit,S) through nn_idx / nn_w (B,L,3)
//          (captra_three_nn_weights): (w0 f[j0] + w1 f[j1]) + w2 f[j2] as captra_interp_concat; y (B,c[nl],L) fp32.
// Outputs equal the layer-by-layer route (captra_dense_bf16_tile_ex chains + captra_gemv_bf16 + captra_interp_concat) bit for bit.
// -2: shapes outside the kernel (channel counts not multiples of 32 / hidden widths beyond 256 then 512 / L % 4 != 0 / ...).
extern "C" int captra_neck_chain_bf16(int kind, int b, long long l, int nl, const int *c, const float *x, const float *x2, int csplit,
                                      const unsigned char *const *wimg, const float *const *bias, const int *nn_idx, const float *nn_w,
                                      int s_known, const float *v, const unsigned short *gw, int cv, int act_last, float *y, captra_stream_t stream) {
    if (kind < 0 || kind > 2 || b < 0 || l < 0 || nl < 2 || nl > 3) return -1;
    if (b == 0 || l == 0) return 0;
    if (l % 4 != 0 || l < 4) return -2;
    for (int i = 1; i <= nl; ++i)
        if (c[i] % 32 != 0 || c[i] < 32) return -2;
    if (c[1] > 256 || (nl == 3 && c[2] > 512) || c[1] % 128 != 0 || (nl == 3 && c[2] % 128 != 0)) return -2;   // images in LDS, whole 8-k-step chunks
    if (csplit < 0 || csplit > c[0]) return -1;
    if (kind == 1 && (csplit != c[0] || c[1] > 256 || c[1] % 4 != 0 || v == nullptr || gw == nullptr)) return -2;
    if (kind == 2 && (nn_idx == nullptr || nn_w == nullptr || s_known < 1)) return -2;
    if (kind == 0 && act_last != ACT_RELU) return -2;
    NkParams p;
    p.L = l;
    for (int i = 0; i < nl; ++i) {
        p.ly[i].wimg = wimg[i]; p.ly[i].bias = bias[i]; p.ly[i].cin = c[i]; p.ly[i].cout = c[i + 1];
        p.ly[i].kst = (c[i] + 15) / 16; p.ly[i].nt = c[i + 1] / 32;
    }
    if (nl == 2) p.ly[2] = p.ly[1];
    p.x = x; p.x2 = x2; p.csplit = csplit; p.nn_idx = nn_idx; p.nn_w = nn_w; p.s_known = s_known;
    p.v = v; p.gw = gw; p.cv = cv; p.gldw = (c[1] + 127) / 128 * 128; p.y = y; p.act_last = act_last;
    hipStream_t s = (hipStream_t)stream;
    // row tiles per wave: eight waves share a layer's nt row tiles
    const int mt1 = (p.ly[0].nt + 7) / 8, mt2 = (p.ly[1].nt + 7) / 8, mt3 = nl == 3 ? (p.ly[2].nt + 7) / 8 : 0;
    if (kind == 0) {
        if (const int zrc = captra_zero_async(y, (size_t)b * c[nl] * sizeof(float), s)) return zrc;      // (a kernel, not a memset node: common.h)
        if (nl == 3 && mt1 == 1 && mt2 == 2 && mt3 == 4) {
            if (g_nk_split == 1) return nk_launch<0, 1, 1, 2, 4>(b, p, s);
            if (g_nk_split == 2) return nk_launch<0, 1, 1, 2, 2>(b, p, s, 2);
            return nk_launch<0, 1, 1, 2, 1>(b, p, s, 4);
        }
        return -2;
    }
    if (nl != 2 || mt1 != 1 || mt2 != 1) return -2;
    if (kind == 1) return nk_launch<1, 0, 1, 1, 0>(b, p, s);
    return nk_launch<2, 0, 1, 1, 0>(b, p, s);
}
