// Gradients of the two indexed-gather operators without float atomics, for gfx950.
//
//   group_points:       out[c][p] = points[c][idx[p]]                =>  d points[c][n] = sum_{p : idx[p] = n} d out[c][p]
//   three_interpolate:  out[c][n] = sum_j w[n][j] points[c][idx[n][j]]  =>  d points[c][m] = sum_{(n,j) : idx[n][j] = m} w[n][j] d out[c][n]
//
// The reference (group_points_gpu.cu:8-44, interpolate_gpu.cu:192-233) and this library's first version scatter with one
// atomicAdd per (channel, position): 5.3 M float atomics per cloud on the SA2 feature group, serialised on the few hundred
// source addresses, in an order that changes from run to run.  The index list is the same for every channel, so it is
// inverted ONCE per call into a CSR structure (positions grouped by source point: an LDS counting sort per cloud, each
// list then sorted ascending), and every (source point, channel) sums its own list in that fixed order: no atomics,
// bit-reproducible gradients, and the index arithmetic amortised over the channels.
#include "common.h"

namespace {

constexpr int CSR_T = 1024;
constexpr int CSR_MAX_SRC = 16384;    // source points per cloud the LDS counters hold (64 KiB)

// one workgroup per cloud: start (B, n_src + 1), order (B, npos)
__global__ __launch_bounds__(CSR_T) void build_csr_kernel(int n_src, int npos, const int *__restrict__ idx_all,
                                                          int *__restrict__ start_all, int *__restrict__ order_all) {
    extern __shared__ int cnt[];                       // [n_src] counts, then write cursors
    __shared__ int wave_sum[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *idx = idx_all + (size_t)b * npos;
    int *start = start_all + (size_t)b * (n_src + 1);
    int *order = order_all + (size_t)b * npos;
    for (int i = tid; i < n_src; i += CSR_T) cnt[i] = 0;
    __syncthreads();
    for (int p = tid; p < npos; p += CSR_T) atomicAdd(&cnt[idx[p]], 1);
    __syncthreads();
    // exclusive scan: a contiguous chunk of counters per thread, wave scan of the chunk sums, wave totals through LDS
    const int per = (n_src + CSR_T - 1) / CSR_T;
    const int lo = tid * per, hi = (lo + per) < n_src ? (lo + per) : n_src;
    int local = 0;
    for (int i = lo; i < hi; ++i) local += cnt[i];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < wave; ++w) run += wave_sum[w];
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        start[i] = run;
        cnt[i] = run;                                   // becomes the write cursor of list i
        run += c;
    }
    if (tid == CSR_T - 1) start[n_src] = npos;
    __syncthreads();
    for (int p = tid; p < npos; p += CSR_T) order[atomicAdd(&cnt[idx[p]], 1)] = p;
    __syncthreads();                                    // (global writes of this workgroup are visible to it after the barrier)
    // each list ascending: the summation order no longer depends on which thread won which slot
    for (int i = tid; i < n_src; i += CSR_T) {
        const int s = start[i], e = cnt[i];
        for (int a = s + 1; a < e; ++a) {
            const int v = order[a];
            int q = a - 1;
            while (q >= s && order[q] > v) { order[q + 1] = order[q]; --q; }
            order[q + 1] = v;
        }
    }
}

constexpr int SR_CH = 8;   // channels per thread: one walk of a list serves 8 rows

// grid (ceil(n_src / 256), ceil(c / SR_CH), B); INTERP: position p = 3 n + j carries weight[p] * grad_out[c][p / 3]
template <bool INTERP>
__global__ __launch_bounds__(256) void scatter_reduce_kernel(int c, int n_src, int npos, int row_len,
                                                             const float *__restrict__ grad_out,
                                                             const float *__restrict__ weight, const int *__restrict__ start_all,
                                                             const int *__restrict__ order_all, float *__restrict__ grad_points) {
    const int b = blockIdx.z, c0 = blockIdx.y * SR_CH;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_src) return;
    const int *start = start_all + (size_t)b * (n_src + 1);
    const int *order = order_all + (size_t)b * npos;
    const float *w = INTERP ? weight + (size_t)b * npos : nullptr;
    const int nch = (c - c0) < SR_CH ? (c - c0) : SR_CH;
    const float *go = grad_out + ((size_t)b * c + c0) * row_len;
    float acc[SR_CH];
#pragma unroll
    for (int k = 0; k < SR_CH; ++k) acc[k] = 0.f;
    for (int j = start[i]; j < start[i + 1]; ++j) {
        const int p = order[j];
        const int col = INTERP ? p / 3 : p;
        const float wj = INTERP ? w[p] : 1.f;
#pragma unroll
        for (int k = 0; k < SR_CH; ++k)
            if (k < nch) acc[k] += INTERP ? go[(size_t)k * row_len + col] * wj : go[(size_t)k * row_len + col];
    }
    float *gp = grad_points + ((size_t)b * c + c0) * n_src + i;
#pragma unroll
    for (int k = 0; k < SR_CH; ++k)
        if (k < nch) gp[(size_t)k * n_src] += acc[k];   // the caller hands in zeros (or a gradient to accumulate into)
}

// Rows that fit LDS (row_len <= 16384 floats: the SA2 feature groups, every interpolation): a workgroup stages one
// gradient row with coalesced 16-byte loads and the lists read it from LDS — the scattered 4-byte reads of the kernel
// above cost a 64-byte L2 sector each (0.67 ms for the 63 M reads of the SA2 feature group).
// grid (ceil(c / cpb), B), 1024 threads, dynamic LDS = row_len floats.
template <bool INTERP>
__global__ __launch_bounds__(1024) void scatter_reduce_lds_kernel(int c, int n_src, int npos, int row_len, int cpb,
                                                                  const float *__restrict__ grad_out,
                                                                  const float *__restrict__ weight,
                                                                  const int *__restrict__ start_all, const int *__restrict__ order_all,
                                                                  float *__restrict__ grad_points) {
    extern __shared__ __attribute__((aligned(16))) float row[];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int *start = start_all + (size_t)b * (n_src + 1);
    const int *order = order_all + (size_t)b * npos;
    const float *w = INTERP ? weight + (size_t)b * npos : nullptr;
    const int c_end = (blockIdx.x + 1) * cpb < c ? (blockIdx.x + 1) * cpb : c;
    for (int ch = blockIdx.x * cpb; ch < c_end; ++ch) {
        const float *src = grad_out + ((size_t)b * c + ch) * row_len;
        if ((row_len & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            for (int e = tid; e < row_len / 4; e += 1024) reinterpret_cast<float4 *>(row)[e] = reinterpret_cast<const float4 *>(src)[e];
        } else {
            for (int e = tid; e < row_len; e += 1024) row[e] = src[e];
        }
        __syncthreads();
        float *gp = grad_points + ((size_t)b * c + ch) * n_src;
        for (int i = tid; i < n_src; i += 1024) {
            float acc = 0.f;
            for (int j = start[i]; j < start[i + 1]; ++j) {
                const int p = order[j];
                acc += INTERP ? row[p / 3] * w[p] : row[p];
            }
            gp[i] += acc;
        }
        __syncthreads();
    }
}

}  // namespace

// -> 0 on success, -2 when the shape is outside the CSR path (caller falls back to its atomic kernel).
// idx: (B, npos) entries in [0, n_src); grad_out rows have row_len = npos (group) or npos / 3 (interpolation).
// Scratch of the CSR path in bytes (0: the shape is outside it): B x (n_src + 1) row starts + B x npos ordered positions, int32.
size_t captra_scatter_ws_bytes(int b, int c, int n_src, long long npos) {
    if (c < 8 || b < 1 || n_src > CSR_MAX_SRC || npos > (1ll << 30) || n_src < 1 || npos < 1) return 0;
    return (size_t)b * ((size_t)n_src + 1 + (size_t)npos) * sizeof(int);
}

// The caller owns the scratch (include/captra_hip.h: "the caller owns every buffer, scratch included"): `ws` must hold
// captra_scatter_ws_bytes(...) bytes; nothing is allocated here.
int captra_scatter_reduce(bool interp, int b, int c, int n_src, long long npos, const float *grad_out, const float *weight,
                          const int *idx, float *grad_points, void *workspace, size_t workspace_bytes, hipStream_t s) {
    const size_t need = captra_scatter_ws_bytes(b, c, n_src, npos);
    if (need == 0) return -2;
    if (workspace == nullptr || workspace_bytes < need) return (int)hipErrorInvalidValue;
    int *ws = static_cast<int *>(workspace);
    int *start = ws, *order = ws + (size_t)b * (n_src + 1);
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(build_csr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            CSR_MAX_SRC * (int)sizeof(int));
        once.done();
    }
    CAPTRA_LAUNCH("scatter_csr", build_csr_kernel, dim3(b), dim3(CSR_T), (size_t)n_src * sizeof(int), s, n_src, (int)npos, idx,
                  start, order);
    dim3 grid((n_src + 255) / 256, (c + SR_CH - 1) / SR_CH, b);
    const int row_len = interp ? (int)(npos / 3) : (int)npos;
    if (row_len <= 16384 && c >= 8) {
        static CaptraDeviceOnce once_lds;
        if (once_lds.first_use()) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_reduce_lds_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * (int)sizeof(float));
            hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_reduce_lds_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * (int)sizeof(float));
            once_lds.done();
        }
        int cpb = 1;
        while ((long long)((c + cpb - 1) / cpb) * b > 1024 && cpb < 16) cpb *= 2;   // ~2-4 workgroups per CU
        dim3 g2((c + cpb - 1) / cpb, b);
        if (interp) {
            CAPTRA_LAUNCH("scatter_reduce", scatter_reduce_lds_kernel<true>, g2, dim3(1024), (size_t)row_len * sizeof(float), s, c,
                          n_src, (int)npos, row_len, cpb, grad_out, weight, start, order, grad_points);
        } else {
            CAPTRA_LAUNCH("scatter_reduce", scatter_reduce_lds_kernel<false>, g2, dim3(1024), (size_t)row_len * sizeof(float), s, c,
                          n_src, (int)npos, row_len, cpb, grad_out, weight, start, order, grad_points);
        }
    } else if (interp) {
        CAPTRA_LAUNCH("scatter_reduce", scatter_reduce_kernel<true>, grid, dim3(256), 0, s, c, n_src, (int)npos, row_len, grad_out,
                      weight, start, order, grad_points);
    } else {
        CAPTRA_LAUNCH("scatter_reduce", scatter_reduce_kernel<false>, grid, dim3(256), 0, s, c, n_src, (int)npos, row_len, grad_out,
                      weight, start, order, grad_points);
    }
    return captra_last_error();
}
