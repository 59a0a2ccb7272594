// Nearest-neighbour search and 3-point interpolation for gfx950.
//
// Replaces three_nn_kernel_fast (reference interpolate_gpu.cu:81-124), knn_kernel_fast (:9-57),
// three_interpolate_kernel_fast (:149-169) and three_interpolate_grad_kernel_fast (:192-214).
// The reference streams `known` from global memory per thread (12-byte stride) and re-reads
// idx/weight for every channel.  Here `known` is staged into LDS as SoA and read with wave-wide
// broadcasts; interpolation stages a chunk of feature rows in LDS and reads idx/weight once per
// position for all staged channels.
//
// three_nn / knn comparator: the reference compares the fp32 distance against `double best`
// initialised to 1e40.  With fp32 `best` initialised to +inf every comparison has the same
// outcome (any finite d is < both; d = inf or NaN is < neither) and an untouched slot is stored
// as (float)1e40 = +inf in both, so fp32 state reproduces the reference exactly.
#include "common.h"

#include <math.h>

// scatter_reduce.hip
int captra_scatter_reduce(bool interp, int b, int c, int n_src, long long npos, const float *grad_out, const float *weight,
                          const int *idx, float *grad_points, void *workspace, size_t workspace_bytes, hipStream_t s);
size_t captra_scatter_ws_bytes(int b, int c, int n_src, long long npos);

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 4096;  // known points per LDS tile (48 KiB)

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(int n, int m,
                                                              const float *__restrict__ unknown_all,
                                                              const float *__restrict__ known_all,
                                                              float *__restrict__ dist2_all,
                                                              int *__restrict__ idx_all) {
    __shared__ __attribute__((aligned(16))) float xs[NN_TILE];
    __shared__ __attribute__((aligned(16))) float ys[NN_TILE];
    __shared__ __attribute__((aligned(16))) float zs[NN_TILE];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int pt = blockIdx.x * NN_THREADS + tid;
    const float *known = known_all + (size_t)b * m * 3;
    const bool live = pt < n;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (live) {
        const float *u = unknown_all + ((size_t)b * n + pt) * 3;
        ux = u[0];
        uy = u[1];
        uz = u[2];
    }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int t0 = 0; t0 < m; t0 += NN_TILE) {
        const int tn = (m - t0) < NN_TILE ? (m - t0) : NN_TILE;
        if (t0 > 0) __syncthreads();
        for (int e = tid; e < tn * 3; e += NN_THREADS) {
            float v = known[(size_t)t0 * 3 + e];
            int p = e / 3, comp = e - p * 3;
            float *dst = comp == 0 ? xs : (comp == 1 ? ys : zs);
            dst[p] = v;
        }
        __syncthreads();
        for (int k = 0; k < tn; ++k) {
            const float d = dist2_unfused(ux, uy, uz, xs[k], ys[k], zs[k]);
            const int kk = t0 + k;
            if (d < b1) {
                b3 = b2; i3 = i2;
                b2 = b1; i2 = i1;
                b1 = d;  i1 = kk;
            } else if (d < b2) {
                b3 = b2; i3 = i2;
                b2 = d;  i2 = kk;
            } else if (d < b3) {
                b3 = d;  i3 = kk;
            }
        }
    }
    if (live) {
        float *dd = dist2_all + ((size_t)b * n + pt) * 3;
        int *ii = idx_all + ((size_t)b * n + pt) * 3;
        dd[0] = b1; dd[1] = b2; dd[2] = b3;
        ii[0] = i1; ii[1] = i2; ii[2] = i3;
    }
}

// k-nearest (k <= 200): sorted insertion with strict '<', state in per-thread scratch exactly as
// large as the reference's `double best[200]; int besti[200]`.  API completeness: CAPTRA never
// enables knn=True (pointnet_utils.py:192), so this is not tuned.
constexpr int KNN_MAXK = 200;
__global__ __launch_bounds__(NN_THREADS) void knn_kernel(int n, int m, int k,
                                                         const float *__restrict__ unknown_all,
                                                         const float *__restrict__ known_all,
                                                         float *__restrict__ dist2_all,
                                                         int *__restrict__ idx_all) {
    const int b = blockIdx.y;
    const int pt = blockIdx.x * NN_THREADS + threadIdx.x;
    if (pt >= n) return;
    const float *u = unknown_all + ((size_t)b * n + pt) * 3;
    const float *known = known_all + (size_t)b * m * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    float best[KNN_MAXK];
    int besti[KNN_MAXK];
    for (int i = 0; i < k; ++i) {
        best[i] = INFINITY;
        besti[i] = 0;
    }
    for (int i = 0; i < m; ++i) {
        const float d = dist2_unfused(ux, uy, uz, known[(size_t)i * 3], known[(size_t)i * 3 + 1],
                                      known[(size_t)i * 3 + 2]);
        if (!(d < best[k - 1])) continue;  // not better than the current worst: no slot changes
        int j = k - 1;
        // shift worse entries down; stop at the first slot whose value is <= d (strict '<' rule)
        while (j > 0 && d < best[j - 1]) {
            best[j] = best[j - 1];
            besti[j] = besti[j - 1];
            --j;
        }
        best[j] = d;
        besti[j] = i;
    }
    float *dd = dist2_all + ((size_t)b * n + pt) * k;
    int *ii = idx_all + ((size_t)b * n + pt) * k;
    for (int i = 0; i < k; ++i) {
        dd[i] = best[i];
        ii[i] = besti[i];
    }
}

constexpr int TI_THREADS = 256;
constexpr int TI_LDS_BYTES = 64 * 1024;
constexpr int TI_POS_PER_BLOCK = 2048;

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_kernel(int c, int m, int n, int cc,
                                                                       const float *__restrict__ points,
                                                                       const int *__restrict__ idx,
                                                                       const float *__restrict__ weight,
                                                                       float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * cc;
    const int ccv = (c - c0) < cc ? (c - c0) : cc;
    const int p0 = blockIdx.x * TI_POS_PER_BLOCK;
    const int p1 = (p0 + TI_POS_PER_BLOCK) < n ? (p0 + TI_POS_PER_BLOCK) : n;
    const int tid = threadIdx.x;
    const float *src = points + ((size_t)b * c + c0) * m;
    const size_t nstage = (size_t)ccv * m;
    for (size_t e = tid; e < nstage; e += TI_THREADS) rows[e] = src[e];
    __syncthreads();
    for (int p = p0 + tid; p < p1; p += TI_THREADS) {
        const int *id = idx + ((size_t)b * n + p) * 3;
        const float *w = weight + ((size_t)b * n + p) * 3;
        const int j0 = id[0], j1 = id[1], j2 = id[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        for (int ch = 0; ch < ccv; ++ch) {
            const float *row = rows + (size_t)ch * m;
            out[((size_t)b * c + c0 + ch) * n + p] = (w0 * row[j0] + w1 * row[j1]) + w2 * row[j2];
        }
    }
}

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_direct_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
    const int b = blockIdx.z;
    const int p = blockIdx.x * TI_THREADS + threadIdx.x;
    if (p >= n) return;
    const int *id = idx + ((size_t)b * n + p) * 3;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int j0 = id[0], j1 = id[1], j2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int ch = blockIdx.y; ch < c; ch += gridDim.y) {
        const float *row = points + ((size_t)b * c + ch) * m;
        out[((size_t)b * c + ch) * n + p] = (w0 * row[j0] + w1 * row[j1]) + w2 * row[j2];
    }
}

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int p = blockIdx.x * TI_THREADS + threadIdx.x;
    if (p >= n) return;
    const int *id = idx + ((size_t)b * n + p) * 3;
    const float *w = weight + ((size_t)b * n + p) * 3;
    const int j0 = id[0], j1 = id[1], j2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int ch = blockIdx.y; ch < c; ch += gridDim.y) {
        const float g = grad_out[((size_t)b * c + ch) * n + p];
        float *row = grad_points + ((size_t)b * c + ch) * m;
        atomicAdd(row + j0, g * w0);
        atomicAdd(row + j1, g * w1);
        atomicAdd(row + j2, g * w2);
    }
}

}  // namespace

extern "C" int captra_three_nn(int b, int n, int m, const float *unknown, const float *known,
                               float *dist2, int *idx, captra_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return -1;
    if (b == 0 || n == 0) return 0;
    dim3 grid((n + NN_THREADS - 1) / NN_THREADS, b);
    CAPTRA_LAUNCH("three_nn", three_nn_kernel, grid, dim3(NN_THREADS), 0, (hipStream_t)stream, n, m, unknown,
                  known, dist2, idx);
    return captra_last_error();
}

extern "C" int captra_knn(int b, int n, int m, int k, const float *unknown, const float *known,
                          float *dist2, int *idx, captra_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || k < 1 || k > KNN_MAXK) return -1;
    if (b == 0 || n == 0) return 0;
    dim3 grid((n + NN_THREADS - 1) / NN_THREADS, b);
    CAPTRA_LAUNCH("knn", knn_kernel, grid, dim3(NN_THREADS), 0, (hipStream_t)stream, n, m, k, unknown, known,
                  dist2, idx);
    return captra_last_error();
}

extern "C" int captra_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                        const float *weight, float *out, captra_stream_t stream) {
    if (b < 0 || c < 0 || m < 0 || n < 0) return -1;
    if (b == 0 || c == 0 || n == 0) return 0;
    if (m == 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = (size_t)m * sizeof(float);
    if (row_bytes > (size_t)TI_LDS_BYTES) {
        dim3 grid((n + TI_THREADS - 1) / TI_THREADS, c < 64 ? c : 64, b);
        CAPTRA_LAUNCH("three_interpolate", three_interpolate_direct_kernel, grid, dim3(TI_THREADS), 0, s, c, m, n,
                      points, idx, weight, out);
        return captra_last_error();
    }
    int cc = (int)(TI_LDS_BYTES / row_bytes);
    if (cc > c) cc = c;
    if (cc > 32) cc = 32;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(three_interpolate_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, TI_LDS_BYTES);
        once.done();
    }
    dim3 grid((n + TI_POS_PER_BLOCK - 1) / TI_POS_PER_BLOCK, (c + cc - 1) / cc, b);
    CAPTRA_LAUNCH("three_interpolate", three_interpolate_kernel, grid, dim3(TI_THREADS), (size_t)cc * row_bytes, s,
                  c, m, n, cc, points, idx, weight, out);
    return captra_last_error();
}

// workspace == nullptr: float atomics like interpolate_gpu.cu:192-214 (no scratch).  With the caller's scratch the positions
// (n, j) are grouped by the known point they read and summed per point in ascending order (scatter_reduce.hip).
static int launch_interp_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                              float *grad_points, void *workspace, size_t workspace_bytes, captra_stream_t stream) {
    if (b < 0 || c < 0 || m < 0 || n < 0) return -1;
    if (b == 0 || c == 0 || n == 0) return 0;
    if (workspace != nullptr) {
        const int rc = captra_scatter_reduce(true, b, c, m, 3ll * n, grad_out, weight, idx, grad_points, workspace, workspace_bytes,
                                             (hipStream_t)stream);
        if (rc != -2) return rc;
    }
    dim3 grid((n + TI_THREADS - 1) / TI_THREADS, c < 64 ? c : 64, b);
    CAPTRA_LAUNCH("three_interpolate_grad", three_interpolate_grad_kernel, grid, dim3(TI_THREADS), 0,
                  (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    return captra_last_error();
}

extern "C" int captra_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                             const int *idx, const float *weight, float *grad_points,
                                             captra_stream_t stream) {
    return launch_interp_grad(b, c, n, m, grad_out, idx, weight, grad_points, nullptr, 0, stream);
}

extern "C" size_t captra_three_interpolate_grad_ws_bytes(int b, int c, int n, int m) {
    return captra_scatter_ws_bytes(b, c, m, 3ll * n);
}

extern "C" int captra_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                                const float *weight, float *grad_points, void *workspace,
                                                size_t workspace_bytes, captra_stream_t stream) {
    return launch_interp_grad(b, c, n, m, grad_out, idx, weight, grad_points, workspace, workspace_bytes, stream);
}
