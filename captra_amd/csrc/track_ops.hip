// Elementwise / small fused operators of the tracking path for gfx950:
//   captra_canonicalize            networks.py:38-41, 184-187
//   captra_three_nn_weights / captra_interp_concat / captra_fp_interpolate_concat
//                                  pointnet_utils.py:280-294 (three_nn + weights + interpolate + cat)
//   captra_group_norm_relu         blocks.py:70-71 (GroupNorm(C/2, C)) + ReLU of MLPConv1d
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
                                                           float *__restrict__ out_n3,
                                                           float *__restrict__ out_planes) {
    const int q = blockIdx.y;  // cloud index b*P + part
    const int bi = q / p;
    const int i = blockIdx.x * 256 + threadIdx.x;
    // out_planes: the cloud a third time, in the ball query's LDS plane order (csrc/bq_scan.h; slots beyond n hold +inf)
    const int npad = (n + 255) & ~255;
    const int pa = ((((i >> 6) >> 2) << 6) + (i & 63)) * 4 + ((i >> 6) & 3);
    if (i >= n) {
        if (out_planes != nullptr && i < npad) {
#pragma unroll
            for (int a = 0; a < 3; ++a) out_planes[((size_t)q * 3 + a) * npad + pa] = __builtin_inff();
        }
        return;
    }
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
        if (out_planes) out_planes[((size_t)q * 3 + a) * npad + pa] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// feature-propagation input, two kernels so that the geometric half can be shared by networks that
// see the same cloud (CoordNet and the root part's RotationNet cloud):
//   three_nn_weights: 3-NN of each unknown point among `known` + weights w_j = (1/(sqrt(d2_j)+1e-8)) / sum
//                     (pointnet_utils.py:284-287 with the CUDA three_nn semantics)
//   interp_concat:    out = cat([skip, sum_j w_j feat_known[:, idx_j]])   (pointnet_utils.py:289-294)
// ---------------------------------------------------------------------------------------------
constexpr int NW_THREADS = 256;
constexpr int NW_TILE = 4096;  // known points per LDS tile (48 KiB)

__global__ __launch_bounds__(NW_THREADS) void three_nn_weights_kernel(int n, int s, const float *__restrict__ unknown,
                                                                      const float *__restrict__ known,
                                                                      int *__restrict__ idx, float *__restrict__ weight) {
    __shared__ __attribute__((aligned(16))) float xs[NW_TILE];
    __shared__ __attribute__((aligned(16))) float ys[NW_TILE];
    __shared__ __attribute__((aligned(16))) float zs[NW_TILE];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int pt = blockIdx.x * NW_THREADS + tid;
    const bool live = pt < n;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (live) {
        const float *u = unknown + ((size_t)b * n + pt) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    const float *kn = known + (size_t)b * s * 3;
    for (int t0 = 0; t0 < s; t0 += NW_TILE) {
        const int tn = (s - t0) < NW_TILE ? (s - t0) : NW_TILE;
        if (t0 > 0) __syncthreads();
        for (int e = tid; e < tn * 3; e += NW_THREADS) {
            const float v = kn[(size_t)t0 * 3 + e];
            const int pp = e / 3, comp = e - pp * 3;
            (comp == 0 ? xs : (comp == 1 ? ys : zs))[pp] = v;
        }
        __syncthreads();
        // branch-free insertion into the sorted triple (strict '<': an equal distance keeps the earlier index), four known
        // points per 16-byte LDS read; the tail of a tile that is not a multiple of 4 is read as +inf padding below
        const int tn4 = (tn + 3) & ~3;
        for (int e = tn + tid; e < tn4; e += NW_THREADS) { xs[e] = INFINITY; ys[e] = INFINITY; zs[e] = INFINITY; }
        if (tn4 != tn) __syncthreads();
        for (int k = 0; k < tn4; k += 4) {
            const float4 kx = *reinterpret_cast<const float4 *>(xs + k);
            const float4 ky = *reinterpret_cast<const float4 *>(ys + k);
            const float4 kz = *reinterpret_cast<const float4 *>(zs + k);
            const float qx[4] = {kx.x, kx.y, kx.z, kx.w}, qy[4] = {ky.x, ky.y, ky.z, ky.w}, qz[4] = {kz.x, kz.y, kz.z, kz.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = dist2_unfused(ux, uy, uz, qx[u], qy[u], qz[u]);   // padding: inf - never inserted (NaN-free: inf*inf)
                const int kk = t0 + k + u;
                const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
                b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? kk : i3);
                b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? kk : i2);
                b1 = c1 ? d : b1;              i1 = c1 ? kk : i1;
            }
        }
    }
    if (live) {
        const float r0 = 1.0f / (sqrtf(b1) + 1e-8f);
        const float r1 = 1.0f / (sqrtf(b2) + 1e-8f);
        const float r2 = 1.0f / (sqrtf(b3) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        int *ii = idx + ((size_t)b * n + pt) * 3;
        float *ww = weight + ((size_t)b * n + pt) * 3;
        ii[0] = i1; ii[1] = i2; ii[2] = i3;
        ww[0] = r0 / norm; ww[1] = r1 / norm; ww[2] = r2 / norm;
    }
}

// The same search with FOUR lanes per unknown point.  One lane per point is a chain of s dependent insertions (distance ->
// three compares -> ten selects per known point): 512 known points are ~150 cycles each with nothing to overlap them when the
// launch has one wave per SIMD (4096 unknowns x 16 clouds), and the kernel sits on the serial prefix of every frame (geometry
// before both networks).  Here lane q of a point's four takes the groups of four known points g = q, q + 4, ... (adjacent lanes
// read adjacent 16-byte LDS words: no bank conflict), keeps its own sorted triple, and the four triples are merged through
// lane shuffles by (distance, index) -- the three smallest in that order are exactly what the sequential strict-'<' scan keeps,
// so indices and weights are bit-identical.
__global__ __launch_bounds__(NW_THREADS) void three_nn_weights4_kernel(int n, int s, const float *__restrict__ unknown,
                                                                       const float *__restrict__ known,
                                                                       int *__restrict__ idx, float *__restrict__ weight) {
    __shared__ __attribute__((aligned(16))) float xs[NW_TILE];
    __shared__ __attribute__((aligned(16))) float ys[NW_TILE];
    __shared__ __attribute__((aligned(16))) float zs[NW_TILE];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, q = tid & 3;
    const int pt = blockIdx.x * (NW_THREADS / 4) + (tid >> 2);
    const bool live = pt < n;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (live) {
        const float *u = unknown + ((size_t)b * n + pt) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
    const float *kn = known + (size_t)b * s * 3;
    for (int t0 = 0; t0 < s; t0 += NW_TILE) {
        const int tn = (s - t0) < NW_TILE ? (s - t0) : NW_TILE;
        if (t0 > 0) __syncthreads();
        for (int e = tid; e < tn * 3; e += NW_THREADS) {
            const float v = kn[(size_t)t0 * 3 + e];
            const int pp = e / 3, comp = e - pp * 3;
            (comp == 0 ? xs : (comp == 1 ? ys : zs))[pp] = v;
        }
        const int tn4 = (tn + 3) & ~3;
        for (int e = tn + tid; e < tn4; e += NW_THREADS) { xs[e] = INFINITY; ys[e] = INFINITY; zs[e] = INFINITY; }
        __syncthreads();
        for (int k = 4 * q; k < tn4; k += 16) {
            const float4 kx = *reinterpret_cast<const float4 *>(xs + k);
            const float4 ky = *reinterpret_cast<const float4 *>(ys + k);
            const float4 kz = *reinterpret_cast<const float4 *>(zs + k);
            const float qx[4] = {kx.x, kx.y, kx.z, kx.w}, qy[4] = {ky.x, ky.y, ky.z, ky.w}, qz[4] = {kz.x, kz.y, kz.z, kz.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = dist2_unfused(ux, uy, uz, qx[u], qy[u], qz[u]);   // padding: inf - never inserted
                const int kk = t0 + k + u;
                const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;                 // (a lane's own indices ascend: strict '<')
                b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? kk : i3);
                b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? kk : i2);
                b1 = c1 ? d : b1;              i1 = c1 ? kk : i1;
            }
        }
    }
    // merge the other three lanes' triples: order by (distance, index)
    const int lane0 = (tid & 63) & ~3;
    float m1 = b1, m2 = b2, m3 = b3;
    int j1 = i1, j2 = i2, j3 = i3;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const int srcl = lane0 + ((q + r) & 3);
        const float od[3] = {__shfl(b1, srcl, 64), __shfl(b2, srcl, 64), __shfl(b3, srcl, 64)};
        const int oi[3] = {__shfl(i1, srcl, 64), __shfl(i2, srcl, 64), __shfl(i3, srcl, 64)};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float d = od[e];
            const int kk = oi[e];
            const bool c1 = d < m1 || (d == m1 && kk < j1), c2 = d < m2 || (d == m2 && kk < j2), c3 = d < m3 || (d == m3 && kk < j3);
            m3 = c2 ? m2 : (c3 ? d : m3);  j3 = c2 ? j2 : (c3 ? kk : j3);
            m2 = c1 ? m1 : (c2 ? d : m2);  j2 = c1 ? j1 : (c2 ? kk : j2);
            m1 = c1 ? d : m1;              j1 = c1 ? kk : j1;
        }
    }
    if (live && q == 0) {
        const float r0 = 1.0f / (sqrtf(m1) + 1e-8f);
        const float r1 = 1.0f / (sqrtf(m2) + 1e-8f);
        const float r2 = 1.0f / (sqrtf(m3) + 1e-8f);
        const float norm = (r0 + r1) + r2;
        int *ii = idx + ((size_t)b * n + pt) * 3;
        float *ww = weight + ((size_t)b * n + pt) * 3;
        // (fewer than three known points: the sequential scan leaves index 0 in the unused slots)
        ii[0] = j1 == 0x7fffffff ? 0 : j1; ii[1] = j2 == 0x7fffffff ? 0 : j2; ii[2] = j3 == 0x7fffffff ? 0 : j3;
        ww[0] = r0 / norm; ww[1] = r1 / norm; ww[2] = r2 / norm;
    }
}

constexpr int IC_THREADS = 256;
constexpr int IC_POS = 1024;           // positions per workgroup (4 per lane)
constexpr int IC_LDS_FLOATS = 8 * 1024;  // 32 KiB of feature rows per workgroup

// grid (pos tiles, channel chunks, B): channel chunk cb covers output channels [cb*cc, cb*cc+cc) of the
// concatenated tensor; channels < c1 are copied from skip, the rest interpolated from LDS-staged rows.
__global__ __launch_bounds__(IC_THREADS) void interp_concat_kernel(int n, int s, int c1, int c2, int cc,
                                                                   const float *__restrict__ skip,
                                                                   const float *__restrict__ feat_known,
                                                                   const int *__restrict__ idx,
                                                                   const float *__restrict__ weight,
                                                                   float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float rows[IC_LDS_FLOATS];
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * IC_POS;
    const int ct = c1 + c2;
    const int ch0 = blockIdx.y * cc;
    const int ch1 = (ch0 + cc) < ct ? (ch0 + cc) : ct;
    // interpolated part of this chunk: feature channels [f0, f1)
    const int f0 = (ch0 > c1 ? ch0 : c1) - c1, f1 = (ch1 > c1 ? ch1 : c1) - c1;
    if (f1 > f0) {
        const float *src = feat_known + ((size_t)b * c2 + f0) * s;
        const size_t nstage = (size_t)(f1 - f0) * s;
        for (size_t e = tid; e < nstage; e += IC_THREADS) rows[e] = src[e];
    }
    __syncthreads();
    int j0[4], j1[4], j2[4];
    float w0[4], w1[4], w2[4];
    // a lane owns FOUR CONSECUTIVE positions when the rows allow 16-byte accesses (n % 4 == 0, aligned tensors): its twelve
    // indices / weights are three 16-byte loads each and every output row segment one 16-byte store (a wave writes 1 KiB
    // contiguous per channel instead of four 256-byte pieces 1 KiB apart); otherwise positions tid, tid + 256, ...
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip) |
                                       reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(weight)) & 15) == 0;
    const int pbase = vec ? p0 + tid * 4 : p0 + tid, pstep = vec ? 1 : IC_THREADS;
    if (vec && pbase + 3 < n) {
        const int4 *ii = reinterpret_cast<const int4 *>(idx + ((size_t)b * n + pbase) * 3);
        const float4 *ww = reinterpret_cast<const float4 *>(weight + ((size_t)b * n + pbase) * 3);
        const int4 a0 = ii[0], a1 = ii[1], a2 = ii[2];
        const float4 b0 = ww[0], b1 = ww[1], b2 = ww[2];
        j0[0] = a0.x; j1[0] = a0.y; j2[0] = a0.z; j0[1] = a0.w; j1[1] = a1.x; j2[1] = a1.y;
        j0[2] = a1.z; j1[2] = a1.w; j2[2] = a2.x; j0[3] = a2.y; j1[3] = a2.z; j2[3] = a2.w;
        w0[0] = b0.x; w1[0] = b0.y; w2[0] = b0.z; w0[1] = b0.w; w1[1] = b1.x; w2[1] = b1.y;
        w0[2] = b1.z; w1[2] = b1.w; w2[2] = b2.x; w0[3] = b2.y; w1[3] = b2.z; w2[3] = b2.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int pt = pbase + u * pstep;
            if (pt >= n) pt = n - 1;
            const int *ii = idx + ((size_t)b * n + pt) * 3;
            const float *ww = weight + ((size_t)b * n + pt) * 3;
            j0[u] = ii[0]; j1[u] = ii[1]; j2[u] = ii[2];
            w0[u] = ww[0]; w1[u] = ww[1]; w2[u] = ww[2];
        }
    }
    for (int ch = ch0; ch < ch1; ++ch) {
        float *orow = out + ((size_t)b * ct + ch) * n;
        if (vec) {
            if (pbase >= n) continue;            // (n % 4 == 0: a lane's four positions are all inside or all outside)
            float4 v;
            if (ch < c1) {
                v = *reinterpret_cast<const float4 *>(skip + ((size_t)b * c1 + ch) * n + pbase);
            } else {
                const float *row = rows + (size_t)(ch - c1 - f0) * s;
                v.x = (w0[0] * row[j0[0]] + w1[0] * row[j1[0]]) + w2[0] * row[j2[0]];
                v.y = (w0[1] * row[j0[1]] + w1[1] * row[j1[1]]) + w2[1] * row[j2[1]];
                v.z = (w0[2] * row[j0[2]] + w1[2] * row[j1[2]]) + w2[2] * row[j2[2]];
                v.w = (w0[3] * row[j0[3]] + w1[3] * row[j1[3]]) + w2[3] * row[j2[3]];
            }
            *reinterpret_cast<float4 *>(orow + pbase) = v;
        } else if (ch < c1) {
            const float *srow = skip + ((size_t)b * c1 + ch) * n;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pt = p0 + tid + u * IC_THREADS;
                if (pt < n) orow[pt] = srow[pt];
            }
        } else {
            const float *row = rows + (size_t)(ch - c1 - f0) * s;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pt = p0 + tid + u * IC_THREADS;
                if (pt < n) orow[pt] = (w0[u] * row[j0[u]] + w1[u] * row[j1[u]]) + w2[u] * row[j2[u]];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm (groups of `cpg` consecutive channels) + optional ReLU over (B,C,N), one workgroup per
// (cloud, group): the group's cpg*N values are read once into registers, mean and centred variance
// reduced in the workgroup, normalised values written once.  Replaces torch's three kernels
// (moments, affine, clamp) after the rotation-head convs (blocks.py:70-71, 148-165).
// ---------------------------------------------------------------------------------------------
constexpr int GN_THREADS = 256;
constexpr int GN_MAXV = 16;  // float4 per thread: supports cpg*N <= 256*16*4 = 16384 values per group

__device__ __forceinline__ float block_sum(float v, float *smem) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    return (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

__global__ __launch_bounds__(GN_THREADS) void group_norm_relu_kernel(int c, int n, int cpg, float eps, int relu,
                                                                     const float *__restrict__ x,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta,
                                                                     float *__restrict__ y) {
    __shared__ float smem[4];
    const int groups = c / cpg;
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const size_t base = ((size_t)b * c + (size_t)g * cpg) * n;   // the group's values are contiguous
    const int total = cpg * n;                                      // multiple of 4 (host checks)
    const int nv = total / 4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x + base);
    float4 v[GN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int e = threadIdx.x + i * GN_THREADS;
        v[i] = e < nv ? x4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = block_sum(sum, smem) / (float)total;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int e = threadIdx.x + i * GN_THREADS;
        if (e < nv) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float var = block_sum(sq, smem) / (float)total;
    const float rstd = 1.0f / sqrtf(var + eps);
    float4 *y4 = reinterpret_cast<float4 *>(y + base);
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int e = threadIdx.x + i * GN_THREADS;
        if (e < nv) {
            const int ch = g * cpg + (e * 4) / n;   // n % 4 == 0: a float4 never straddles channels
            const float ga = gamma[ch] * rstd, be = beta[ch];
            float4 o;
            o.x = (v[i].x - mean) * ga + be;
            o.y = (v[i].y - mean) * ga + be;
            o.z = (v[i].z - mean) * ga + be;
            o.w = (v[i].w - mean) * ga + be;
            if (relu) {
                o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
                o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
            }
            y4[e] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CoordinateNet read-out: softmax over the S segmentation logits of every point (networks.py:50, F.softmax(dim=1)) and the
// label = first index of the largest logit (model.py:466, torch.max(seg, dim=-2)[1] on the softmax: the exponential is
// monotone, so the arg max is the logits') in ONE launch -- replaces a softmax, an arg-max reduction and an int64 -> int32
// copy.  S <= 8.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_softmax_argmax_kernel(int s, int n, const float *__restrict__ logits,
                                                                 float *__restrict__ seg, int *__restrict__ labels) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *x = logits + (size_t)b * s * n + i;
    float v[8];
    float m = 0.f;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < s) {
            v[c] = x[(size_t)c * n];
            if (c == 0 || v[c] > m) {      // strict '>': the FIRST index of the maximum, as torch.max / argmax return
                arg = c;
                m = v[c];
            }
        }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < s) {
            v[c] = expf(v[c] - m);
            sum += v[c];
        }
    if (seg != nullptr) {
        float *y = seg + (size_t)b * s * n + i;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < s) y[(size_t)c * n] = v[c] / sum;
    }
    if (labels != nullptr) labels[(size_t)b * n + i] = arg;
}

// ---------------------------------------------------------------------------------------------
// several small device-to-device copies in one launch (the lanes' pose / record hand-over: ten 4-microsecond copy
// kernels per lane and frame otherwise).  Jobs of 4-byte words; blockIdx.y = job.
// ---------------------------------------------------------------------------------------------
constexpr int CM_MAX_JOBS = 16;
struct CopyMulti {
    const unsigned *src[CM_MAX_JOBS];
    unsigned *dst[CM_MAX_JOBS];
    long long words[CM_MAX_JOBS];
};
__global__ __launch_bounds__(256) void copy_multi_kernel(CopyMulti m) {
    const int j = blockIdx.y;
    const long long nw = m.words[j];
    const unsigned *s = m.src[j];
    unsigned *d = m.dst[j];
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const long long n4 = nw >> 2;
        for (long long e = t0; e < n4; e += step) reinterpret_cast<uint4 *>(d)[e] = reinterpret_cast<const uint4 *>(s)[e];
        for (long long e = (n4 << 2) + t0; e < nw; e += step) d[e] = s[e];
    } else {
        for (long long e = t0; e < nw; e += step) d[e] = s[e];
    }
}

// out[r] = max over the l floats of row r (group_all pooling of a (B,C,N) tensor: pointnet_utils.py:342 torch.max(new_points, 2)):
// 32 lanes per row, 16-byte loads when the rows allow it
__global__ __launch_bounds__(256) void row_max_kernel(long long rows, int l, const float *__restrict__ x, float *__restrict__ out) {
    const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int lane = threadIdx.x & 31;
    float m = -__builtin_inff();
    if (r < rows) {
        const float *row = x + r * l;
        if ((l & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
            for (int i = lane * 4; i < l; i += 128) {
                const float4 v = *reinterpret_cast<const float4 *>(row + i);
                m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            }
        } else {
            for (int i = lane; i < l; i += 32) m = fmaxf(m, row[i]);
        }
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
    if (r < rows && lane == 0) out[r] = m;
}

// The packed pose record of the per-frame exchange (captra_amd/parallel.py: [R(9) t(3) s(1) valid(1)] per trajectory and part)
__global__ __launch_bounds__(256) void pack_pose_kernel(int n, const float *__restrict__ rot, const float *__restrict__ trans,
                                                        const float *__restrict__ scale, const float *__restrict__ valid,
                                                        float *__restrict__ out0, float *__restrict__ out1) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 14) return;
    const int r = e / 14, f = e - r * 14;
    const float v = f < 9 ? rot[r * 9 + f] : f < 12 ? trans[r * 3 + (f - 9)] : f == 12 ? scale[r] : (valid != nullptr ? valid[r] : 1.f);
    out0[e] = v;
    if (out1 != nullptr) out1[e] = v;
}

}  // namespace

extern "C" int captra_canonicalize_planes(int b, int p, int n, const float *pts, const float *mean, const float *rot,
                                          const float *trans, const float *scale, float *out_cn, float *out_n3,
                                          float *out_planes, captra_stream_t stream) {
    if (b < 0 || p < 1 || n < 0) return -1;
    if (b == 0 || n == 0) return 0;
    dim3 grid((n + 255) / 256, b * p);
    CAPTRA_LAUNCH("canonicalize", canonicalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, n, pts, mean,
                  rot, trans, scale, out_cn, out_n3, out_planes);
    return captra_last_error();
}

extern "C" int captra_canonicalize(int b, int p, int n, const float *pts, const float *mean, const float *rot,
                                   const float *trans, const float *scale, float *out_cn, float *out_n3,
                                   captra_stream_t stream) {
    return captra_canonicalize_planes(b, p, n, pts, mean, rot, trans, scale, out_cn, out_n3, nullptr, stream);
}

static CAPTRA_KNOB int g_nn_split = 1;     // experiment knob: 0 = one lane per unknown point (the first form)
extern "C" void captra_three_nn_set_split(int on) { g_nn_split = on; }

extern "C" int captra_three_nn_weights(int b, int n, int s, const float *unknown, const float *known, int *idx,
                                      float *weight, captra_stream_t stream) {
    if (b < 0 || n < 0 || s < 0) return -1;
    if (b == 0 || n == 0) return 0;
    if (s == 0) return -1;
    if (g_nn_split) {
        // four lanes per unknown point (a quarter of the dependent chain each); the one-lane form when that would only add waves
        // to a chip that is full anyway
        dim3 grid4((n + NW_THREADS / 4 - 1) / (NW_THREADS / 4), b);
        CAPTRA_LAUNCH("three_nn_weights", three_nn_weights4_kernel, grid4, dim3(NW_THREADS), 0, (hipStream_t)stream, n, s,
                      unknown, known, idx, weight);
        return captra_last_error();
    }
    dim3 grid((n + NW_THREADS - 1) / NW_THREADS, b);
    CAPTRA_LAUNCH("three_nn_weights", three_nn_weights_kernel, grid, dim3(NW_THREADS), 0, (hipStream_t)stream, n, s,
                  unknown, known, idx, weight);
    return captra_last_error();
}

extern "C" int captra_interp_concat(int b, int n, int s, int c1, int c2, const float *skip, const float *feat_known,
                                    const int *idx, const float *weight, float *out, captra_stream_t stream) {
    if (b < 0 || n < 0 || s < 1 || c1 < 0 || c2 < 0) return -1;
    if (c1 > 0 && skip == nullptr) return -1;
    if (b == 0 || n == 0 || c1 + c2 == 0) return 0;
    if (s > IC_LDS_FLOATS) return -2;
    int cc = IC_LDS_FLOATS / s;   // channels per chunk: their feature rows fit the LDS buffer
    if (cc > 32) cc = 32;
    dim3 grid((n + IC_POS - 1) / IC_POS, (c1 + c2 + cc - 1) / cc, b);
    CAPTRA_LAUNCH("interp_concat", interp_concat_kernel, grid, dim3(IC_THREADS), 0, (hipStream_t)stream, n, s, c1, c2,
                  cc, skip, feat_known, idx, weight, out);
    return captra_last_error();
}

extern "C" int captra_fp_interpolate_concat(int b, int n, int s, int c1, int c2, const float *unknown,
                                            const float *known, const float *skip, const float *feat_known,
                                            int *idx_scratch, float *weight_scratch, float *out,
                                            captra_stream_t stream) {
    int err = captra_three_nn_weights(b, n, s, unknown, known, idx_scratch, weight_scratch, stream);
    if (err) return err;
    return captra_interp_concat(b, n, s, c1, c2, skip, feat_known, idx_scratch, weight_scratch, out, stream);
}

extern "C" int captra_group_norm_relu(int b, int c, int n, int channels_per_group, float eps, int relu,
                                      const float *x, const float *gamma, const float *beta, float *y,
                                      captra_stream_t stream) {
    if (b < 0 || c < 1 || n < 1 || channels_per_group < 1 || c % channels_per_group != 0) return -1;
    if (n % 4 != 0 || (long long)channels_per_group * n > (long long)GN_THREADS * GN_MAXV * 4) return -2;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return -2;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("group_norm_relu", group_norm_relu_kernel, dim3(b * (c / channels_per_group)), dim3(GN_THREADS), 0,
                  (hipStream_t)stream, c, n, channels_per_group, eps, relu, x, gamma, beta, y);
    return captra_last_error();
}

// logits (B,S,N), S <= 8 -> seg (B,S,N) = softmax over S (may be NULL), labels (B,N) i32 = first index of the largest logit
// (may be NULL): networks.py:50 + model.py:466 in one launch.
extern "C" int captra_seg_softmax_argmax(int b, int s, int n, const float *logits, float *seg, int *labels, captra_stream_t stream) {
    if (b < 0 || s < 1 || n < 0) return -1;
    if (s > 8) return -2;
    if (b == 0 || n == 0) return 0;
    CAPTRA_LAUNCH("seg_readout", seg_softmax_argmax_kernel, dim3((n + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, s, n, logits, seg, labels);
    return captra_last_error();
}

// njobs <= 16 device-to-device copies of bytes[j] (multiples of 4) bytes each, one launch; host arrays of DEVICE pointers.
extern "C" int captra_copy_multi(int njobs, const void *const *src, void *const *dst, const long long *bytes, captra_stream_t stream) {
    if (njobs < 0 || njobs > CM_MAX_JOBS) return -1;
    if (njobs == 0) return 0;
    CopyMulti m;
    long long most = 0;
    for (int j = 0; j < njobs; ++j) {
        if (bytes[j] < 0 || (bytes[j] & 3) != 0 || ((reinterpret_cast<uintptr_t>(src[j]) | reinterpret_cast<uintptr_t>(dst[j])) & 3) != 0) return -1;
        m.src[j] = reinterpret_cast<const unsigned *>(src[j]);
        m.dst[j] = reinterpret_cast<unsigned *>(dst[j]);
        m.words[j] = bytes[j] >> 2;
        most = m.words[j] > most ? m.words[j] : most;
    }
    if (most == 0) return 0;
    long long blocks = (most / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    CAPTRA_LAUNCH("copy_multi", copy_multi_kernel, dim3((unsigned)blocks, njobs), dim3(256), 0, (hipStream_t)stream, m);
    return captra_last_error();
}

// n = B*P pose records [R(9) t(3) s(1) valid(1)] from rot (n,3,3), trans (n,3), scale (n), valid (n) floats (NULL = all 1)
// -> out0 (n,14) and, when non-NULL, a second copy out1: the per-frame exchange's packing in one launch.
extern "C" int captra_pack_pose(int n, const float *rot, const float *trans, const float *scale, const float *valid, float *out0,
                                float *out1, captra_stream_t stream) {
    if (n < 0) return -1;
    if (n == 0) return 0;
    CAPTRA_LAUNCH("pack_pose", pack_pose_kernel, dim3((n * 14 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, rot, trans, scale,
                  valid, out0, out1);
    return captra_last_error();
}

// out (rows) = max over each row of x (rows, l), l >= 1: the pooling of a group_all set abstraction on a (B,C,N) tensor (rows = B*C).
extern "C" int captra_row_max(long long rows, int l, const float *x, float *out, captra_stream_t stream) {
    if (rows < 0 || l < 1) return -1;
    if (rows == 0) return 0;
    CAPTRA_LAUNCH("row_max", row_max_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, rows, l, x, out);
    return captra_last_error();
}
