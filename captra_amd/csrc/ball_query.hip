// Ball query (radius neighbourhood search) for gfx950.
//
// Replaces ball_query_kernel_fast (reference ball_query_gpu.cu:9-45), which runs one THREAD per
// centre streaming the whole cloud from global memory with a 12-byte stride and a divergent
// early exit.  Here:
//   * the cloud is staged once per workgroup into LDS as three coordinate planes and reused by
//     BQ_WAVES * BQ_CPW centres; four 64-point chunks are interleaved per lane so that one
//     ds_read_b128 per plane feeds 256 distance tests, computed with packed fp32 instructions;
//   * one WAVE owns a centre: chunks of 64 consecutive points are tested in order, the hit mask comes
//     from the compare itself (wave64 ballot), the output slot of each hit is
//     cnt + popcount(mask below my lane) — an ordered compaction that reproduces the
//     reference's "first nsample hits in index order" bit for bit — and the early exit is
//     wave-uniform;
//   * up to 4 radii share one scan (the multi-scale grouping of PointNetSetAbstractionMsg,
//     pointnet_utils.py:228-233 asks the same centres for 3 radii).
// Distance: ((cx-x)^2 + (cy-y)^2) + (cz-z)^2, unfused fp32, strict '<' against radius*radius.
#include "common.h"

namespace {

constexpr int BQ_WAVES = 16;        // waves per workgroup (two workgroups per CU hold the full 32 wave slots; M = 512 -> 16 per cloud)
constexpr int BQ_CPW = 2;           // centres per wave
constexpr int BQ_TILE = 8192;       // points staged per LDS tile (96 KiB)
constexpr int BQ_MAXR = 4;

struct BqParams {
    float r2[BQ_MAXR];
    int ns[BQ_MAXR];
    int *idx[BQ_MAXR];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int bq_pad(int v) { return (v + 255) & ~255; }

template <int NR>
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(int n, int m,
                                                                   const float *__restrict__ new_xyz_all,
                                                                   const float *__restrict__ xyz_all,
                                                                   BqParams prm) {
    // three planes (x, y, z); inside a plane the 64-point chunks are interleaved four by four:
    // element ((chunk / 4) * 64 + lane) * 4 + chunk % 4, so a lane fetches its coordinate of four
    // consecutive chunks with ONE ds_read_b128 and the distances of 256 points cost 16 packed-fp32
    // instructions per wave.  Slots past the end of the cloud hold +inf and never hit.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tile_cap = bq_pad(n < BQ_TILE ? n : BQ_TILE);
    float *xs = lds;
    float *ys = xs + tile_cap;
    float *zs = ys + tile_cap;

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float *new_xyz = new_xyz_all + (size_t)b * m * 3;
    const int c_base = (blockIdx.x * BQ_WAVES + wave) * BQ_CPW;
    int cnt[BQ_CPW][NR];
    int first[BQ_CPW][NR];
#pragma unroll
    for (int ci = 0; ci < BQ_CPW; ++ci)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            cnt[ci][r] = 0;
            first[ci][r] = 0;
        }

    for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
        const int tn = (n - t0) < BQ_TILE ? (n - t0) : BQ_TILE;
        const int tn_pad = bq_pad(tn);
        if (t0 > 0) __syncthreads();
        for (int p = tid; p < tn_pad; p += BQ_WAVES * 64) {
            const bool inb = p < tn;
            const float *q = xyz + (size_t)(t0 + (inb ? p : 0)) * 3;
            const float inf = __builtin_inff();
            const int chunk = p >> 6;
            const int a = (((chunk >> 2) << 6) + (p & 63)) * 4 + (chunk & 3);
            xs[a] = inb ? q[0] : inf;
            ys[a] = inb ? q[1] : inf;
            zs[a] = inb ? q[2] : inf;
        }
        __syncthreads();

#pragma unroll
        for (int ci = 0; ci < BQ_CPW; ++ci) {
            const int c = c_base + ci;
            if (c >= m) continue;
            bool open = false;
#pragma unroll
            for (int r = 0; r < NR; ++r) open = open || (cnt[ci][r] < prm.ns[r]);
            if (!open) continue;
            const float cx = new_xyz[(size_t)c * 3 + 0];
            const float cy = new_xyz[(size_t)c * 3 + 1];
            const float cz = new_xyz[(size_t)c * 3 + 2];
            const f32x4 cx4 = {cx, cx, cx, cx}, cy4 = {cy, cy, cy, cy}, cz4 = {cz, cz, cz, cz};
            for (int g = 0; g < (tn_pad >> 8) && open; ++g) {
                const f32x4 dx = cx4 - reinterpret_cast<const f32x4 *>(xs)[g * 64 + lane];
                const f32x4 dy = cy4 - reinterpret_cast<const f32x4 *>(ys)[g * 64 + lane];
                const f32x4 dz = cz4 - reinterpret_cast<const f32x4 *>(zs)[g * 64 + lane];
                const f32x4 d2 = (dx * dx + dy * dy) + dz * dz;  // unfused: the build runs with -ffp-contract=off
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if (!open) break;
                    const int k0 = t0 + g * 256 + h * 64;
                    bool any_open = false;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        if (cnt[ci][r] < prm.ns[r]) {
                            const bool hit = d2[h] < prm.r2[r];
                            const unsigned long long mask = __ballot(hit);
                            if (mask) {
                                // rank of this lane among the hits = hits in lower lanes (v_mbcnt), written through a
                                // uniform row pointer + unsigned 32-bit slot (scalar base + VGPR offset addressing)
                                const unsigned pos = (unsigned)cnt[ci][r] +
                                                     __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                if (cnt[ci][r] == 0) first[ci][r] = k0 + (__ffsll((long long)mask) - 1);
                                int *row = prm.idx[r] + ((size_t)b * m + c) * prm.ns[r];
                                if (hit && pos < (unsigned)prm.ns[r]) row[pos] = k0 + lane;
                                cnt[ci][r] += __popcll(mask);
                            }
                            any_open = any_open || (cnt[ci][r] < prm.ns[r]);
                        }
                    }
                    open = any_open;
                }
            }
        }
    }

    // pad the tail of every list with the first hit (0 when the ball is empty)
#pragma unroll
    for (int ci = 0; ci < BQ_CPW; ++ci) {
        const int c = c_base + ci;
        if (c >= m) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int have = cnt[ci][r] < prm.ns[r] ? cnt[ci][r] : prm.ns[r];
            const int fill = first[ci][r];
            int *row = prm.idx[r] + ((size_t)b * m + c) * prm.ns[r];
            for (int s = have + lane; s < prm.ns[r]; s += 64) row[s] = fill;
        }
    }
}

int launch_ball_query(int b, int n, int m, int nr, const float *radius, const int *nsample,
                      const float *new_xyz, const float *xyz, int *const *idx, hipStream_t s) {
    if (b < 0 || n < 0 || m < 0 || nr < 1 || nr > BQ_MAXR) return -1;
    if (b == 0 || m == 0) return 0;
    BqParams prm;
    for (int r = 0; r < BQ_MAXR; ++r) {
        prm.r2[r] = 0.f;
        prm.ns[r] = 0;
        prm.idx[r] = nullptr;
    }
    for (int r = 0; r < nr; ++r) {
        if (nsample[r] < 0) return -1;
        prm.r2[r] = radius[r] * radius[r];
        prm.ns[r] = nsample[r];
        prm.idx[r] = idx[r];
    }
    const int tile_cap = bq_pad(n < BQ_TILE ? n : BQ_TILE);
    size_t shmem = (size_t)(tile_cap > 0 ? tile_cap : 256) * 3 * sizeof(float);
    dim3 grid((m + BQ_WAVES * BQ_CPW - 1) / (BQ_WAVES * BQ_CPW), b);
    dim3 block(BQ_WAVES * 64);
#define BQ_LAUNCH(NR)                                                                            \
    {                                                                                            \
        auto kern = ball_query_kernel<NR>;                                                       \
        static CaptraDeviceOnce once;                                                            \
        if (once.first_use())                                                                    \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, BQ_TILE * 12);       \
        CAPTRA_LAUNCH("ball_query", kern, grid, block, shmem, s, n, m, new_xyz, xyz, prm);       \
    }
    switch (nr) {
        case 1: BQ_LAUNCH(1) break;
        case 2: BQ_LAUNCH(2) break;
        case 3: BQ_LAUNCH(3) break;
        default: BQ_LAUNCH(4) break;
    }
#undef BQ_LAUNCH
    return captra_last_error();
}

}  // namespace

extern "C" int captra_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                 const float *xyz, int *idx, captra_stream_t stream) {
    int *idxs[1] = {idx};
    return launch_ball_query(b, n, m, 1, &radius, &nsample, new_xyz, xyz, idxs, (hipStream_t)stream);
}

extern "C" int captra_ball_query_multi(int b, int n, int m, int nr, const float *radius,
                                       const int *nsample, const float *new_xyz, const float *xyz,
                                       int *const *idx, captra_stream_t stream) {
    return launch_ball_query(b, n, m, nr, radius, nsample, new_xyz, xyz, idx, (hipStream_t)stream);
}
