// Ball query (radius neighbourhood search) for gfx950.
//
// Replaces ball_query_kernel_fast (reference ball_query_gpu.cu:9-45), which runs one THREAD per
// centre streaming the whole cloud from global memory with a 12-byte stride and a divergent
// early exit.  Here:
//   * the cloud is staged once per workgroup into LDS as SoA (x[], y[], z[]) with coalesced
//     global reads, and reused by CENTRES_PER_BLOCK centres;
//   * one WAVE owns a centre: 64 consecutive points are tested per step, the hit mask comes
//     from the compare itself (wave64 ballot), the output slot of each hit is
//     cnt + popcount(mask below my lane) — an ordered compaction that reproduces the
//     reference's "first nsample hits in index order" bit for bit — and the early exit is
//     wave-uniform;
//   * up to 4 radii share one scan (the multi-scale grouping of PointNetSetAbstractionMsg,
//     pointnet_utils.py:228-233 asks the same centres for 3 radii).
// Distance: ((cx-x)^2 + (cy-y)^2) + (cz-z)^2, unfused fp32, strict '<' against radius*radius.
#include "common.h"

namespace {

constexpr int BQ_WAVES = 8;         // waves per workgroup
constexpr int BQ_CPW = 2;           // centres per wave
constexpr int BQ_TILE = 8192;       // points staged per LDS tile (96 KiB)
constexpr int BQ_MAXR = 4;

struct BqParams {
    float r2[BQ_MAXR];
    int ns[BQ_MAXR];
    int *idx[BQ_MAXR];
};

template <int NR>
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(int n, int m,
                                                                   const float *__restrict__ new_xyz_all,
                                                                   const float *__restrict__ xyz_all,
                                                                   BqParams prm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tile_cap = n < BQ_TILE ? n : BQ_TILE;
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
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

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
        if (t0 > 0) __syncthreads();
        // coalesced AoS read -> SoA LDS
        for (int e = tid; e < tn * 3; e += BQ_WAVES * 64) {
            float v = xyz[(size_t)t0 * 3 + e];
            int p = e / 3, comp = e - p * 3;
            float *dst = comp == 0 ? xs : (comp == 1 ? ys : zs);
            dst[p] = v;
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
            for (int k0 = 0; k0 < tn; k0 += 64) {
                const int kl = k0 + lane;
                const bool inb = kl < tn;
                const int ks = inb ? kl : 0;
                const float d2 = dist2_unfused(cx, cy, cz, xs[ks], ys[ks], zs[ks]);
                bool any_open = false;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (cnt[ci][r] < prm.ns[r]) {
                        const bool hit = inb && (d2 < prm.r2[r]);
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            const int pos = cnt[ci][r] + __popcll(mask & lt_mask);
                            if (cnt[ci][r] == 0) first[ci][r] = t0 + k0 + (__ffsll((long long)mask) - 1);
                            if (hit && pos < prm.ns[r])
                                prm.idx[r][((size_t)b * m + c) * prm.ns[r] + pos] = t0 + kl;
                            cnt[ci][r] += __popcll(mask);
                        }
                        any_open = any_open || (cnt[ci][r] < prm.ns[r]);
                    }
                }
                if (!any_open) break;
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
    const int tile_cap = n < BQ_TILE ? n : BQ_TILE;
    size_t shmem = (size_t)(tile_cap > 0 ? tile_cap : 1) * 3 * sizeof(float);
    dim3 grid((m + BQ_WAVES * BQ_CPW - 1) / (BQ_WAVES * BQ_CPW), b);
    dim3 block(BQ_WAVES * 64);
#define BQ_LAUNCH(NR)                                                                            \
    {                                                                                            \
        auto kern = ball_query_kernel<NR>;                                                       \
        static bool attr_set = false;                                                            \
        if (!attr_set) {                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, BQ_TILE * 12);       \
            attr_set = true;                                                                     \
        }                                                                                        \
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
