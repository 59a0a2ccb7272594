// Grouping / gathering (indexed row copies) and their scatter-add gradients for gfx950.
//
// Replaces group_points_kernel_fast (reference group_points_gpu.cu:47-66) and
// gather_points_kernel_fast (sampling_gpu.cu:8-24), which launch one thread per output element
// over a (positions, C, B) grid and re-read idx for every channel.  This is a pure data-movement
// op bounded by the HBM write of out (4*C*M*K bytes): here a workgroup stages a chunk of
// `cc` channel rows of points[b] (contiguous cc*N floats) into LDS with 16-byte coalesced loads,
// reads each idx quad ONCE, gathers the cc rows from LDS and writes 16-byte coalesced stores.
#include "common.h"
#include "bq_scan.h"

// scatter_reduce.hip: CSR inversion of an index list + per-source-point sums (-2: shape outside that path)
int captra_scatter_reduce(bool interp, int b, int c, int n_src, long long npos, const float *grad_out, const float *weight,
                          const int *idx, float *grad_points, void *workspace, size_t workspace_bytes, hipStream_t s);
size_t captra_scatter_ws_bytes(int b, int c, int n_src, long long npos);

namespace {

constexpr int GP_THREADS = 256;
constexpr int GP_LDS_BYTES = 64 * 1024;  // channel-chunk staging budget (2 workgroups / CU)
constexpr int GP_POS_PER_BLOCK = 4096;   // output positions handled by one workgroup
// measurement knobs (tools/bench_group.py --sweep): staging budget in KiB (<= 64), channel-chunk cap, positions per workgroup
CAPTRA_KNOB int g_gp_lds_kb = 64, g_gp_ccmax = 32, g_gp_ppb = 0;

// VEC = 4: npos % 4 == 0 and 16-byte aligned idx/out rows; VEC = 1 otherwise.
// Channel-outer order: a thread keeps the idx quads of its positions in registers and the workgroup
// finishes one channel's contiguous run of positions (16 KiB per sweep) before moving to the next
// row — measured 5.8 TB/s on the SA2 feature groups against 4.8 TB/s for the position-outer order
// (32 interleaved 1-KiB streams per wave), of the 5.9-6.3 TB/s a plain fill reaches on this part
// (tools/ubench/hbm_probe.hip).  Streaming (non-temporal) stores: written once, never re-read here.
template <int VEC>
__device__ __forceinline__ void group_points_body(int c, int n, long long npos, int cc, int ppb, const float *__restrict__ points,
                                                  const int *__restrict__ idx, float *__restrict__ out, int bx, int by, int b) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int c0 = by * cc;
    const int ccv = (c - c0) < cc ? (c - c0) : cc;
    const long long p0 = (long long)bx * ppb;
    const long long p1 = (p0 + ppb) < npos ? (p0 + ppb) : npos;
    const int tid = threadIdx.x;

    const float *src = points + ((size_t)b * c + c0) * n;
    const size_t nstage = (size_t)ccv * n;
    if ((nstage & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(rows);
        for (size_t e = tid; e < nstage / 4; e += GP_THREADS) d4[e] = s4[e];
    } else {
        for (size_t e = tid; e < nstage; e += GP_THREADS) rows[e] = src[e];
    }
    __syncthreads();

    const int *idx_b = idx + (size_t)b * npos;
    float *out_b = out + ((size_t)b * c + c0) * npos;
    if (VEC == 4) {
        constexpr int Q = 4;  // quads per thread per sweep: 4096 positions
        for (long long pb = p0; pb < p1; pb += GP_THREADS * 4 * Q) {
            int4 id[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const long long p = pb + ((long long)q * GP_THREADS + tid) * 4;
                id[q] = p < p1 ? *reinterpret_cast<const int4 *>(idx_b + p) : make_int4(0, 0, 0, 0);
            }
            for (int ch = 0; ch < ccv; ++ch) {
                const float *row = rows + (size_t)ch * n;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const long long p = pb + ((long long)q * GP_THREADS + tid) * 4;
                    const f32x4 vv = {row[id[q].x], row[id[q].y], row[id[q].z], row[id[q].w]};
                    if (p < p1)
                        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(out_b + (size_t)ch * npos + p));
                }
            }
        }
    } else {
        for (long long p = p0 + tid; p < p1; p += GP_THREADS) {
            const int id = idx_b[p];
            for (int ch = 0; ch < ccv; ++ch) out_b[(size_t)ch * npos + p] = rows[(size_t)ch * n + id];
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(GP_THREADS) void group_points_kernel(int c, int n, long long npos, int cc, int ppb,
                                                                  const float *__restrict__ points,
                                                                  const int *__restrict__ idx,
                                                                  float *__restrict__ out) {
    group_points_body<VEC>(c, n, npos, cc, ppb, points, idx, out, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several grouping jobs on the SAME source clouds' size (one set-abstraction level: every radius x every feature tensor of
// both networks) in ONE launch: the 10-35 MB jobs of a level are per-launch-latency bound one at a time (seventeen launches
// per frame, ~0.09 ms of ramps and tails in a 0.46 ms job).  blockIdx.x walks the jobs' (position block, channel chunk)
// grids back to back; each workgroup runs the single-job body on its job's descriptor.
constexpr int GP_MAX_JOBS = 12;
struct GpJob {
    const float *points;
    const int *idx;
    float *out;
    long long npos;
    int c, cc, ppb, pos_blocks, blk0;
};
struct GpMulti {
    GpJob job[GP_MAX_JOBS];
    int njobs, n;
};

__global__ __launch_bounds__(GP_THREADS) void group_points_multi_kernel(GpMulti m) {
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < m.njobs; ++k)
        if ((int)blockIdx.x >= m.job[k].blk0) j = k;
    const GpJob &g = m.job[j];
    const int local = (int)blockIdx.x - g.blk0;
    group_points_body<4>(g.c, m.n, g.npos, g.cc, g.ppb, g.points, g.idx, g.out, local % g.pos_blocks, local / g.pos_blocks, blockIdx.z);
}

// Rows too long for LDS staging (n*4 > budget): gather straight from global / L2.
__global__ __launch_bounds__(GP_THREADS) void group_points_direct_kernel(int c, int n, long long npos,
                                                                         const float *__restrict__ points,
                                                                         const int *__restrict__ idx,
                                                                         float *__restrict__ out) {
    const int b = blockIdx.z;
    const long long p = (long long)blockIdx.x * GP_THREADS + threadIdx.x;
    if (p >= npos) return;
    const int id = idx[(size_t)b * npos + p];
    const float *src = points + (size_t)b * c * n;
    float *dst = out + (size_t)b * c * npos;
    for (int ch = blockIdx.y; ch < c; ch += gridDim.y) dst[(size_t)ch * npos + p] = src[(size_t)ch * n + id];
}

__global__ __launch_bounds__(GP_THREADS) void group_points_grad_kernel(int c, int n, long long npos,
                                                                       const float *__restrict__ grad_out,
                                                                       const int *__restrict__ idx,
                                                                       float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const long long p = (long long)blockIdx.x * GP_THREADS + threadIdx.x;
    if (p >= npos) return;
    const int id = idx[(size_t)b * npos + p];
    for (int ch = blockIdx.y; ch < c; ch += gridDim.y)
        atomicAdd(grad_points + ((size_t)b * c + ch) * n + id, grad_out[((size_t)b * c + ch) * npos + p]);
}


// ======================================================================================================================
// QueryAndGroup in ONE launch (reference pointnet_lib/pointnet2_utils.py:274-310: ball_query -> grouping_operation(xyz) -> centre
// subtraction -> grouping_operation(features) -> cat([features, xyz])).  A workgroup owns MCB consecutive centres of a cloud and a
// chunk of the output's channels: it stages the cloud in LDS as the ball query's coordinate planes, its waves scan for their
// centres four at a time (bq_scan.h: the same index-order scan as captra_ball_query), the lists stay in LDS -- and are written out
// once when the caller wants them --, then the workgroup writes its channels position-major exactly as group_points_body does
// (16-byte streaming stores, a channel's MCB x K positions contiguous): feature channels gathered from LDS-staged rows (or from
// L2 when a row chunk does not fit), the three coordinate channels straight from the planes minus the centre.  Neither the index
// list nor the grouped coordinates make a round trip through HBM between two launches.
// ======================================================================================================================
struct QgParams {
    int n, m, k, c, use_xyz, cc, cs, mcb, stage_rows;      // cc: channels staged at a time, cs: channels per workgroup (a multiple of cc)
    float r2;
    const float *xyz_n3, *new_xyz, *feat;
    float *out;
    int *idx_out;
};

__device__ __forceinline__ int qg_plane_slot(int p) { return ((p >> 8) << 8) + ((p & 63) << 2) + ((p >> 6) & 3); }

template <int NT>
__global__ __launch_bounds__(NT) void query_and_group_kernel(QgParams q) {
    extern __shared__ __attribute__((aligned(16))) float qg_lds[];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int npad = bq_pad(q.n);
    float *xs = qg_lds, *ys = xs + npad, *zs = ys + npad;
    int *lists = reinterpret_cast<int *>(zs + npad);            // [mcb][k]
    float *ctr = reinterpret_cast<float *>(lists + q.mcb * q.k);   // [mcb][3]
    float *rows = ctr + ((q.mcb * 3 + 3) & ~3);                 // [cc][n] (stage_rows)
    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.x * q.mcb;
    const int ncen = (q.m - c0) < q.mcb ? (q.m - c0) : q.mcb;
    const int ct = (q.feat != nullptr ? q.c : 0) + (q.use_xyz || q.feat == nullptr ? 3 : 0);   // channels of the output
    const int cf = q.feat != nullptr ? q.c : 0;                 // of which features (first)
    const int cs0 = blockIdx.y * q.cs;
    const int cs1 = (cs0 + q.cs) < ct ? (cs0 + q.cs) : ct;      // this workgroup's channels [cs0, cs1), cc at a time
    bq_stage_tile(q.xyz_n3 + (size_t)b * q.n * 3, 0, q.n, xs, ys, zs, tid, NT);
    for (int e = tid; e < ncen * 3; e += NT) ctr[e] = q.new_xyz[((size_t)b * q.m + c0) * 3 + e];
    __syncthreads();
    // ---- the ball query of this workgroup's centres: wave w takes every (NT / 64)-th group of four ----------------------------
    {
        const float r2[1] = {q.r2};
        const int ns[1] = {q.k};
        for (int g0 = 4 * wave; g0 < ncen; g0 += NT / 16) {
            float cx[4], cy[4], cz[4];
            int cnt[4][1], first[4][1];
            int *rws[1] = {lists + (size_t)g0 * q.k};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = g0 + u < ncen;
                const int cl = live ? g0 + u : g0;
                cx[u] = ctr[3 * cl + 0]; cy[u] = ctr[3 * cl + 1]; cz[u] = ctr[3 * cl + 2];
                cnt[u][0] = live ? 0 : q.k;
                first[u][0] = 0;
            }
            bq_scan_centres<4, 1>(xs, ys, zs, npad >> 8, 0, cx, cy, cz, r2, ns, rws, cnt, first, lane);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (g0 + u < ncen) bq_pad_row(rws[0] + (size_t)u * q.k, cnt[u][0], first[u][0], q.k, lane);
        }
    }
    __syncthreads();
    const int npos = ncen * q.k;                                // this workgroup's positions (k % 4 == 0)
    if (q.idx_out != nullptr && blockIdx.y == 0) {
        int4 *dst = reinterpret_cast<int4 *>(q.idx_out + ((size_t)b * q.m + c0) * q.k);
        const int4 *src = reinterpret_cast<const int4 *>(lists);
        for (int e = tid; e < npos / 4; e += NT) dst[e] = src[e];
    }
    // ---- the workgroup's channels, cc at a time: stage the chunk's feature rows, then channel after channel over the positions ---
    const long long mk = (long long)q.m * q.k;
    float *out_b = q.out + (size_t)b * ct * mk + (size_t)c0 * q.k;
    for (int ch0 = cs0; ch0 < cs1; ch0 += q.cc) {
        const int ch1 = (ch0 + q.cc) < cs1 ? (ch0 + q.cc) : cs1;
        const int f1 = ch1 < cf ? ch1 : cf;                     // feature channels [ch0, f1) of this chunk
        if (q.stage_rows && f1 > ch0) {
            if (ch0 > cs0) __syncthreads();                     // (the last chunk's rows have been read)
            const float *src = q.feat + ((size_t)b * q.c + ch0) * q.n;
            const size_t nstage = (size_t)(f1 - ch0) * q.n;
            if ((nstage & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                const float4 *s4 = reinterpret_cast<const float4 *>(src);
                float4 *d4 = reinterpret_cast<float4 *>(rows);
                for (size_t e = tid; e < nstage / 4; e += NT) d4[e] = s4[e];
            } else {
                for (size_t e = tid; e < nstage; e += NT) rows[e] = src[e];
            }
            __syncthreads();
        }
        for (int pb = 0; pb < npos; pb += NT * 16) {
            int4 id[4];
            int cen[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pb + (u * NT + tid) * 4;
                id[u] = p < npos ? *reinterpret_cast<const int4 *>(lists + p) : make_int4(0, 0, 0, 0);
                cen[u] = p < npos ? p / q.k : 0;                // (a quad never straddles two centres: k % 4 == 0)
            }
            for (int ch = ch0; ch < ch1; ++ch) {
                float *orow = out_b + (size_t)ch * mk;
                if (ch < cf) {
                    const float *row = q.stage_rows ? rows + (size_t)(ch - ch0) * q.n : q.feat + ((size_t)b * q.c + ch) * q.n;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb + (u * NT + tid) * 4;
                        const f32x4 vv = {row[id[u].x], row[id[u].y], row[id[u].z], row[id[u].w]};
                        if (p < npos) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(orow + p));
                    }
                } else {
                    const float *plane = xs + (size_t)(ch - cf) * npad;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb + (u * NT + tid) * 4;
                        const float cc0 = ctr[3 * cen[u] + (ch - cf)];
                        const f32x4 vv = {plane[qg_plane_slot(id[u].x)] - cc0, plane[qg_plane_slot(id[u].y)] - cc0,
                                          plane[qg_plane_slot(id[u].z)] - cc0, plane[qg_plane_slot(id[u].w)] - cc0};
                        if (p < npos) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(orow + p));
                    }
                }
            }
        }
    }
}

CAPTRA_KNOB int g_qg_mcb = 0, g_qg_cc = 0, g_qg_cs = 0, g_qg_nt = 0;     // measurement knobs: centres per workgroup, channels per staging chunk / per workgroup, threads (0 = heuristic)

// channel-chunk size, positions per workgroup and position blocks of one job (rows fit the LDS budget)
void gp_shape(int b, int c, int n, long long npos, int &cc, int &ppb, long long &pos_blocks) {
    const size_t row_bytes = (size_t)n * sizeof(float);
    cc = (int)((size_t)g_gp_lds_kb * 1024 / row_bytes);
    if (cc < 1) cc = 1;
    if (cc > c) cc = c;
    if (cc > g_gp_ccmax) cc = g_gp_ccmax;
    // long rows (SA1: 16 KiB each): amortise the staging over twice the positions
    ppb = n >= 2048 ? 2 * GP_POS_PER_BLOCK : GP_POS_PER_BLOCK;
    if (g_gp_ppb > 0) ppb = g_gp_ppb;
    pos_blocks = (npos + ppb - 1) / ppb;
    // few-channel inputs: split the channels over more workgroups until the chip is covered (staging cost per
    // output byte is n / ppb whatever cc is; only the idx quads are re-read)
    while (cc > 1 && pos_blocks * ((c + cc - 1) / cc) * b < 1024) cc = (cc + 1) / 2;
}

int launch_group(int b, int c, int n, long long npos, const float *points, const int *idx, float *out,
                 hipStream_t s) {
    if (b < 0 || c < 0 || n < 0 || npos < 0) return -1;
    if (b == 0 || c == 0 || npos == 0) return 0;
    if (n == 0) return -1;
    const size_t row_bytes = (size_t)n * sizeof(float);
    if (row_bytes > (size_t)GP_LDS_BYTES) {
        dim3 grid((unsigned)((npos + GP_THREADS - 1) / GP_THREADS), c < 64 ? c : 64, b);
        CAPTRA_LAUNCH("group_points", group_points_direct_kernel, grid, dim3(GP_THREADS), 0, s, c, n, npos,
                      points, idx, out);
        return captra_last_error();
    }
    int cc, ppb;
    long long pos_blocks;
    gp_shape(b, c, n, npos, cc, ppb, pos_blocks);
    dim3 grid((unsigned)pos_blocks, (c + cc - 1) / cc, b);
    size_t shmem = (size_t)cc * row_bytes;
    const bool vec = (npos % 4 == 0) && ((reinterpret_cast<uintptr_t>(idx) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(group_points_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void *>(group_points_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES);
        once.done();
    }
    if (vec) {
        CAPTRA_LAUNCH("group_points", group_points_kernel<4>, grid, dim3(GP_THREADS), shmem, s, c, n, npos, cc, ppb,
                      points, idx, out);
    } else {
        CAPTRA_LAUNCH("group_points", group_points_kernel<1>, grid, dim3(GP_THREADS), shmem, s, c, n, npos, cc, ppb,
                      points, idx, out);
    }
    return captra_last_error();
}

// workspace == nullptr: the reference's algorithm (float atomics, group_points_gpu.cu:8-25), which needs no scratch.
// With the caller's scratch (captra_group_points_grad_ws_bytes): the index list, shared by every channel, is inverted once
// and every source point sums its own list in ascending position order (scatter_reduce.hip) -- no atomics, bit-reproducible.
int launch_group_grad(int b, int c, int n, long long npos, const float *grad_out, const int *idx,
                      float *grad_points, void *workspace, size_t workspace_bytes, hipStream_t s) {
    if (b < 0 || c < 0 || n < 0 || npos < 0) return -1;
    if (b == 0 || c == 0 || npos == 0) return 0;
    if (workspace != nullptr) {
        const int rc = captra_scatter_reduce(false, b, c, n, npos, grad_out, nullptr, idx, grad_points, workspace, workspace_bytes, s);
        if (rc != -2) return rc;
    }
    dim3 grid((unsigned)((npos + GP_THREADS - 1) / GP_THREADS), c < 64 ? c : 64, b);
    CAPTRA_LAUNCH("group_points_grad", group_points_grad_kernel, grid, dim3(GP_THREADS), 0, s, c, n, npos,
                  grad_out, idx, grad_points);
    return captra_last_error();
}

}  // namespace

extern "C" void captra_query_and_group_set_shape(int mcb, int cc) { g_qg_mcb = mcb & 0xFF; g_qg_nt = (mcb >> 8) & 0xFFF; g_qg_cc = cc & 0xFF; g_qg_cs = (cc >> 8) & 0xFFF; }   // (bits 8..: threads / channels per workgroup)

// QueryAndGroup(radius, nsample, use_xyz)(xyz, new_xyz, features) of the reference (pointnet2_utils.py:274-310) in one launch:
// out (B, C + 3, M, K) = cat([features[:, :, idx], xyz[idx] - new_xyz]) (features first; (B,3,M,K) when features == NULL; (B,C,M,K)
// when use_xyz == 0), idx = the ball query's lists (also written to idx_out (B,M,K) when non-NULL).  xyz (B,N,3), new_xyz (B,M,3),
// features (B,C,N) or NULL.  -2: nsample % 4 != 0, unaligned out / idx_out, or a cloud whose planes do not fit the LDS (N > 8192):
// the caller runs captra_ball_query + captra_group_points then.
extern "C" int captra_query_and_group(int b, int n, int m, float radius, int nsample, int c, int use_xyz, const float *xyz, const float *new_xyz,
                                      const float *features, float *out, int *idx_out, captra_stream_t stream) {
    if (b < 0 || n < 1 || m < 0 || nsample < 1 || c < 0) return -1;
    if (features == nullptr && !use_xyz) return -1;
    if (b == 0 || m == 0) return 0;
    if (nsample % 4 || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(idx_out) & 15) || n > 8192) return -2;
    if ((long long)m * nsample * (c + 3) >= (1ll << 31)) return -2;
    QgParams q;
    q.n = n; q.m = m; q.k = nsample; q.c = features != nullptr ? c : 0; q.use_xyz = use_xyz; q.r2 = radius * radius;
    q.xyz_n3 = xyz; q.new_xyz = new_xyz; q.feat = features; q.out = out; q.idx_out = idx_out;
    const int ct = q.c + ((use_xyz || features == nullptr) ? 3 : 0);
    // Threads: a long scan with few channels behind it (SA1: 4096 points, <= 6 channels) is bound by the search -- 512 threads, so that
    // two workgroups keep 16 waves on a CU; a short scan with hundreds of channels (SA2) by the stores -- 256 threads as group_points.
    // Centres per workgroup: 4096 positions at most (16 KiB of lists).  Feature rows staged in LDS cc at a time when a chunk of at
    // least 8 fits beside the planes and the lists in 64 KiB (two workgroups per CU), else gathered from L2; a workgroup takes cs
    // channels (a multiple of cc), i.e. repeats the search ceil(ct / cs) times per centre block: as few as still cover the chip.
    int nt = (n >= 2048 && ct <= 16) ? 512 : 256;
    if (g_qg_nt == 256 || g_qg_nt == 512) nt = g_qg_nt;
    // (sweep on the CAPTRA shapes at 32 clouds, tools/bench_qg.py --sweep: SA1 calls 45 -> 34-40 us with 512 threads and 32 centres per
    // workgroup; SA2 K = 64: 32 centres, rows 8 at a time, 48 channels per workgroup 78-82 us (one search per 8 channels: 88-106);
    // K = 128: 16 centres, 8 channels per workgroup 124 us = 5.7 TB/s (48 and more: 140-150))
    const bool many = ct > 16;
    int mcb = (many ? 2048 : 4096) / nsample;
    if (mcb > 32) mcb = 32;
    mcb = (mcb + 3) & ~3;
    if (mcb < 4) mcb = 4;
    if (g_qg_mcb > 0) mcb = (g_qg_mcb + 3) & ~3;
    if (mcb * nsample > 8192) return -2;
    const size_t fixed = (size_t)3 * bq_pad(n) * 4 + (size_t)mcb * nsample * 4 + (size_t)((mcb * 3 + 3) & ~3) * 4;
    int cc = ct;
    q.stage_rows = 0;
    if (q.c > 0) {
        const long long room = 64 * 1024 - (long long)fixed;
        const int fit = room > 0 ? (int)(room / ((long long)n * 4)) : 0;
        if (fit >= 8 || fit >= q.c) { q.stage_rows = 1; cc = fit < 8 ? fit : 8; }
        else cc = 8;
    }
    if (g_qg_cc > 0) cc = g_qg_cc;
    if (cc > ct) cc = ct;
    int cs = !many ? ct : (nsample <= 64 ? 6 * cc : cc);
    if (g_qg_cs > 0) cs = (g_qg_cs + cc - 1) / cc * cc;
    if (cs < cc) cs = cc;
    q.mcb = mcb; q.cc = cc; q.cs = cs;
    const size_t lds = fixed + (q.stage_rows ? (size_t)cc * n * 4 : 0);
    if (lds > 160 * 1024) return -2;
    dim3 grid((m + mcb - 1) / mcb, (ct + cs - 1) / cs, b);
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(query_and_group_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return (int)hipGetLastError();
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(query_and_group_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    if (nt == 512) { CAPTRA_LAUNCH("query_and_group", query_and_group_kernel<512>, grid, dim3(512), lds, (hipStream_t)stream, q); }
    else { CAPTRA_LAUNCH("query_and_group", query_and_group_kernel<256>, grid, dim3(256), lds, (hipStream_t)stream, q); }
    return captra_last_error();
}

extern "C" void captra_group_set_shape(int lds_kb, int ccmax, int ppb) {
    g_gp_lds_kb = lds_kb < 1 ? 1 : (lds_kb > 64 ? 64 : lds_kb);
    g_gp_ccmax = ccmax < 1 ? 1 : ccmax;
    g_gp_ppb = ppb > 0 ? (ppb + 4095) / 4096 * 4096 : 0;
}

extern "C" int captra_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                   const int *idx, float *out, captra_stream_t stream) {
    return launch_group(b, c, n, (long long)npoints * nsample, points, idx, out, (hipStream_t)stream);
}

extern "C" int captra_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                        const float *grad_out, const int *idx, float *grad_points,
                                        captra_stream_t stream) {
    return launch_group_grad(b, c, n, (long long)npoints * nsample, grad_out, idx, grad_points, nullptr, 0,
                             (hipStream_t)stream);
}

extern "C" size_t captra_group_points_grad_ws_bytes(int b, int c, int n, int npoints, int nsample) {
    return captra_scatter_ws_bytes(b, c, n, (long long)npoints * nsample);
}

extern "C" int captra_group_points_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                           const int *idx, float *grad_points, void *workspace, size_t workspace_bytes,
                                           captra_stream_t stream) {
    return launch_group_grad(b, c, n, (long long)npoints * nsample, grad_out, idx, grad_points, workspace, workspace_bytes,
                             (hipStream_t)stream);
}

// gather = group with one sample per centre (out (B,C,npoints))
extern "C" int captra_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                                    float *out, captra_stream_t stream) {
    return launch_group(b, c, n, (long long)npoints, points, idx, out, (hipStream_t)stream);
}

extern "C" int captra_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                         const int *idx, float *grad_points, captra_stream_t stream) {
    return launch_group_grad(b, c, n, (long long)npoints, grad_out, idx, grad_points, nullptr, 0, (hipStream_t)stream);
}

// caller-owned scratch, as captra_group_points_grad_ws: the atomic-free, bit-reproducible path
extern "C" size_t captra_gather_points_grad_ws_bytes(int b, int c, int n, int npoints) {
    return captra_scatter_ws_bytes(b, c, n, (long long)npoints);
}

extern "C" int captra_gather_points_grad_ws(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                            float *grad_points, void *workspace, size_t workspace_bytes, captra_stream_t stream) {
    return launch_group_grad(b, c, n, (long long)npoints, grad_out, idx, grad_points, workspace, workspace_bytes, (hipStream_t)stream);
}

// njobs grouping jobs over clouds of the same size n (host arrays of length njobs: channel counts, centres, samples per centre,
// DEVICE pointers): out[j] (B,c[j],npoints[j],nsample[j]) = points[j] (B,c[j],N) gathered through idx[j] (B,npoints[j],nsample[j]).
// One launch when every job has 16-byte aligned idx / out rows with npoints*nsample % 4 == 0, the rows fit the LDS staging and
// njobs <= 12; otherwise the jobs are launched one by one (same results).
extern "C" int captra_group_points_multi(int b, int n, int njobs, const int *c, const int *npoints, const int *nsample,
                                         const float *const *points, const int *const *idx, float *const *out, captra_stream_t stream) {
    if (b < 0 || n < 1 || njobs < 0) return -1;
    if (b == 0 || njobs == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    bool one = njobs <= GP_MAX_JOBS && (size_t)n * sizeof(float) <= (size_t)GP_LDS_BYTES;
    for (int j = 0; j < njobs && one; ++j) {
        const long long npos = (long long)npoints[j] * nsample[j];
        one = c[j] >= 1 && npos >= 1 && npos % 4 == 0 && (reinterpret_cast<uintptr_t>(idx[j]) & 15) == 0 && (reinterpret_cast<uintptr_t>(out[j]) & 15) == 0;
    }
    if (!one) {
        for (int j = 0; j < njobs; ++j) {
            const int rc = launch_group(b, c[j], n, (long long)npoints[j] * nsample[j], points[j], idx[j], out[j], s);
            if (rc != 0) return rc;
        }
        return 0;
    }
    GpMulti m;
    m.njobs = njobs;
    m.n = n;
    int blocks = 0, ccmax = 0;
    for (int j = 0; j < njobs; ++j) {
        GpJob &g = m.job[j];
        long long pb;
        g.points = points[j]; g.idx = idx[j]; g.out = out[j]; g.c = c[j]; g.npos = (long long)npoints[j] * nsample[j];
        gp_shape(b, g.c, n, g.npos, g.cc, g.ppb, pb);
        g.pos_blocks = (int)pb;
        g.blk0 = blocks;
        blocks += (int)pb * ((g.c + g.cc - 1) / g.cc);
        ccmax = g.cc > ccmax ? g.cc : ccmax;
    }
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(group_points_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES);
        once.done();
    }
    CAPTRA_LAUNCH("group_points", group_points_multi_kernel, dim3(blocks, 1, b), dim3(GP_THREADS), (size_t)ccmax * n * sizeof(float), s, m);
    return captra_last_error();
}
