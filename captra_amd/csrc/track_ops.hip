// Elementwise / small fused operators of the tracking path for gfx950:
//   captra_canonicalize            networks.py:38-41, 184-187
//   captra_fp_interpolate_concat   pointnet_utils.py:280-294 (three_nn + weights + interpolate + cat)
//   captra_group_norm_relu         blocks.py:70-71 (GroupNorm(C/2, C)) + ReLU of MLPConv1d
//   captra_rot_head_pool           blocks.py:183-192 + networks.py:127-138 (per-point rotation
//                                  representation, masked mean over the part's points)
#include "common.h"

#include <math.h>

namespace {

// ---------------------------------------------------------------------------------------------
// canonicalise: out = R^T ((pts + mean) - t) / s, 3-term dot product summed left to right
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void canonicalize_kernel(int p, int n, const float *__restrict__ pts,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ rot,
                                                           const float *__restrict__ trans,
                                                           const float *__restrict__ scale,
                                                           float *__restrict__ out_cn,
                                                           float *__restrict__ out_n3) {
    const int q = blockIdx.y;  // cloud index b*P + part
    const int bi = q / p;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *R = rot + (size_t)q * 9;
    const float *t = trans + (size_t)q * 3;
    const float s = scale[q];
    float v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = (pts[((size_t)bi * 3 + a) * n + i] + mean[bi * 3 + a]) - t[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float acc = (R[0 * 3 + a] * v[0] + R[1 * 3 + a] * v[1]) + R[2 * 3 + a] * v[2];
        const float o = acc / s;
        if (out_cn) out_cn[((size_t)q * 3 + a) * n + i] = o;
        if (out_n3) out_n3[((size_t)q * n + i) * 3 + a] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// feature-propagation input: 3-NN of each unknown point among `known`, weights
// w_j = (1/(sqrt(d2_j)+1e-8)) / sum, out = cat([skip, sum_j w_j feat_known[:, idx_j]])
// One workgroup = 1024 unknown points (4 per thread); `known` then feature rows staged in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int FP_THREADS = 256;
constexpr int FP_PPT = 4;
constexpr int FP_LDS_FLOATS = 16 * 1024;  // 64 KiB, shared by the xyz tile and the feature-row chunk
constexpr int FP_XYZ_TILE = FP_LDS_FLOATS / 3;

__global__ __launch_bounds__(FP_THREADS) void fp_interp_concat_kernel(int n, int s, int c1, int c2, int cc,
                                                                      const float *__restrict__ unknown,
                                                                      const float *__restrict__ known,
                                                                      const float *__restrict__ skip,
                                                                      const float *__restrict__ feat_known,
                                                                      float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[FP_LDS_FLOATS];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * (FP_THREADS * FP_PPT);
    const int ct = c1 + c2;

    float ux[FP_PPT], uy[FP_PPT], uz[FP_PPT];
    float b1[FP_PPT], b2[FP_PPT], b3[FP_PPT];
    int i1[FP_PPT], i2[FP_PPT], i3[FP_PPT];
#pragma unroll
    for (int j = 0; j < FP_PPT; ++j) {
        const int pt = p0 + tid + j * FP_THREADS;
        ux[j] = uy[j] = uz[j] = 0.f;
        if (pt < n) {
            const float *u = unknown + ((size_t)b * n + pt) * 3;
            ux[j] = u[0]; uy[j] = u[1]; uz[j] = u[2];
        }
        b1[j] = b2[j] = b3[j] = INFINITY;
        i1[j] = i2[j] = i3[j] = 0;
    }
    const float *kn = known + (size_t)b * s * 3;
    for (int t0 = 0; t0 < s; t0 += FP_XYZ_TILE) {
        const int tn = (s - t0) < FP_XYZ_TILE ? (s - t0) : FP_XYZ_TILE;
        float *xs = lds, *ys = lds + FP_XYZ_TILE, *zs = lds + 2 * FP_XYZ_TILE;
        if (t0 > 0) __syncthreads();
        for (int e = tid; e < tn * 3; e += FP_THREADS) {
            const float v = kn[(size_t)t0 * 3 + e];
            const int pp = e / 3, comp = e - pp * 3;
            (comp == 0 ? xs : (comp == 1 ? ys : zs))[pp] = v;
        }
        __syncthreads();
        for (int k = 0; k < tn; ++k) {
            const float kx = xs[k], ky = ys[k], kz = zs[k];
            const int kk = t0 + k;
#pragma unroll
            for (int j = 0; j < FP_PPT; ++j) {
                const float d = dist2_unfused(ux[j], uy[j], uz[j], kx, ky, kz);
                if (d < b1[j]) {
                    b3[j] = b2[j]; i3[j] = i2[j];
                    b2[j] = b1[j]; i2[j] = i1[j];
                    b1[j] = d;     i1[j] = kk;
                } else if (d < b2[j]) {
                    b3[j] = b2[j]; i3[j] = i2[j];
                    b2[j] = d;     i2[j] = kk;
                } else if (d < b3[j]) {
                    b3[j] = d;     i3[j] = kk;
                }
            }
        }
    }
    float w1[FP_PPT], w2[FP_PPT], w3[FP_PPT];
#pragma unroll
    for (int j = 0; j < FP_PPT; ++j) {
        const float r0 = 1.0f / (sqrtf(b1[j]) + 1e-8f);
        const float r1 = 1.0f / (sqrtf(b2[j]) + 1e-8f);
        const float r2 = 1.0f / (sqrtf(b3[j]) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        w1[j] = r0 / norm; w2[j] = r1 / norm; w3[j] = r2 / norm;
    }
    // skip connection copied into channels [0, c1)
    for (int ch = 0; ch < c1; ++ch) {
#pragma unroll
        for (int j = 0; j < FP_PPT; ++j) {
            const int pt = p0 + tid + j * FP_THREADS;
            if (pt < n) out[((size_t)b * ct + ch) * n + pt] = skip[((size_t)b * c1 + ch) * n + pt];
        }
    }
    // interpolated features into channels [c1, c1+c2)
    const float *fk = feat_known + (size_t)b * c2 * s;
    if (cc > 0) {
        for (int ch0 = 0; ch0 < c2; ch0 += cc) {
            const int ccv = (c2 - ch0) < cc ? (c2 - ch0) : cc;
            __syncthreads();
            const size_t nstage = (size_t)ccv * s;
            for (size_t e = tid; e < nstage; e += FP_THREADS) lds[e] = fk[(size_t)ch0 * s + e];
            __syncthreads();
            for (int ch = 0; ch < ccv; ++ch) {
                const float *row = lds + (size_t)ch * s;
#pragma unroll
                for (int j = 0; j < FP_PPT; ++j) {
                    const int pt = p0 + tid + j * FP_THREADS;
                    if (pt < n)
                        out[((size_t)b * ct + c1 + ch0 + ch) * n + pt] =
                            (w1[j] * row[i1[j]] + w2[j] * row[i2[j]]) + w3[j] * row[i3[j]];
                }
            }
        }
    } else {
        for (int ch = 0; ch < c2; ++ch) {
            const float *row = fk + (size_t)ch * s;
#pragma unroll
            for (int j = 0; j < FP_PPT; ++j) {
                const int pt = p0 + tid + j * FP_THREADS;
                if (pt < n)
                    out[((size_t)b * ct + c1 + ch) * n + pt] =
                        (w1[j] * row[i1[j]] + w2[j] * row[i2[j]]) + w3[j] * row[i3[j]];
            }
        }
    }
}

}  // namespace

extern "C" int captra_canonicalize(int b, int p, int n, const float *pts, const float *mean, const float *rot,
                                   const float *trans, const float *scale, float *out_cn, float *out_n3,
                                   captra_stream_t stream) {
    if (b < 0 || p < 1 || n < 0) return -1;
    if (b == 0 || n == 0) return 0;
    dim3 grid((n + 255) / 256, b * p);
    CAPTRA_LAUNCH("canonicalize", canonicalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, n, pts, mean,
                  rot, trans, scale, out_cn, out_n3);
    return captra_last_error();
}

extern "C" int captra_fp_interpolate_concat(int b, int n, int s, int c1, int c2, const float *unknown,
                                            const float *known, const float *skip, const float *feat_known,
                                            float *out, captra_stream_t stream) {
    if (b < 0 || n < 0 || s < 0 || c1 < 0 || c2 < 0) return -1;
    if (c1 > 0 && skip == nullptr) return -1;
    if (b == 0 || n == 0) return 0;
    if (s == 0) return -1;
    int cc = FP_LDS_FLOATS / s;
    if (cc > 32) cc = 32;
    dim3 grid((n + FP_THREADS * FP_PPT - 1) / (FP_THREADS * FP_PPT), b);
    CAPTRA_LAUNCH("fp_interpolate_concat", fp_interp_concat_kernel, grid, dim3(FP_THREADS), 0, (hipStream_t)stream,
                  n, s, c1, c2, cc, unknown, known, skip, feat_known, out);
    return captra_last_error();
}
