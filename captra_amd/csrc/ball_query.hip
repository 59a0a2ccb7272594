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
//   * OPT-IN (`captra_ball_query_set_prune(1)`, kernel template `PRUNE`; off by default — measured slower, see below): small
//     radii answered from a cell grid instead of the scan.  A radius whose ball rarely holds nsample points never lets the
//     scan stop early — at SA1's r = 0.05 / 0.1 every centre walks all 4096 points for 20 / 90 hits.  For clouds of
//     1024..4096 points the workgroup also sorts the cloud by grid cell in LDS (cell edge a hair above the largest such
//     radius, so a ball lies inside its centre's 3 x 3 x 3 cells; <= 4096 cells), a wave tests only the points of those 27
//     cells (9 contiguous runs of the sorted copy) with the SAME distance expression, marks every hit in a bitmap over
//     ORIGINAL indices (ds_or) and reads the first nsample set bits back in order — the reference's "first nsample hits
//     in index order", bit for bit, whatever order the candidates were visited in (tests: grid == scan == oracle).  Large
//     radii (r > 0.15 of the cloud's extent: their balls fill within a fraction of the cloud) keep the index-order scan.
//     Measured at 32 clouds (tools/bench_ball_query.py): three radii 89.7 us against the scan's 73.4, r = 0.05 alone 38.8
//     against 31.5.  The grid query itself is faster (about 20 us of the 38.8), but the sorted copy, cell table and bitmaps
//     take 152 KB of LDS — one workgroup per CU instead of two, the sixteen workgroups of a cloud each rebuild the grid —
//     and that costs 19 us (r = 0.2 alone, never answered from the grid: 47.6 against 28.4).  Paying off needs the grid built
//     once per cloud by its own launch into a caller-provided workspace (an ABI addition) and candidates read from L2.
// Distance: ((cx-x)^2 + (cy-y)^2) + (cz-z)^2, unfused fp32, strict '<' against radius*radius.
#include "common.h"
#include "bq_scan.h"

namespace {

constexpr int BQ_WAVES = 16;        // waves per workgroup (two workgroups per CU hold the full 32 wave slots; M = 512 -> 16 per cloud)
constexpr int BQ_CPW = 2;           // centres per wave
constexpr int BQ_TILE = 8192;       // points staged per LDS tile (96 KiB)
constexpr int BQ_MAXR = 4;

constexpr int BQ_PRUNE_MAXN = 4096; // pruned path: largest cloud whose sorted copy + bitmaps fit in LDS beside the planes
constexpr int BQ_PRUNE_MINN = 1024; // below this the scan is 16 chunks at most: no grid
constexpr int BQ_MAXCELLS = 4096;
constexpr int BQ_BMW = BQ_PRUNE_MAXN / 32;   // bitmap words per (wave, radius)

struct BqParams {
    float r2[BQ_MAXR];
    float rad[BQ_MAXR];
    int ns[BQ_MAXR];
    int *idx[BQ_MAXR];
    int m0, mhi;          // centres [m0, mhi) of every cloud (captra_set_centre_window; default 0, m)
};


template <int NR, bool PRUNE, int CPW = BQ_CPW>
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
    // PRUNE only (n <= BQ_PRUNE_MAXN, one tile): sorted copy (x, y, z, original index), cell starts, bitmaps, scratch
    float4 *srt = reinterpret_cast<float4 *>(zs + tile_cap);
    unsigned *cstart = reinterpret_cast<unsigned *>(srt + (PRUNE ? tile_cap : 0));           // [BQ_MAXCELLS + 1]
    unsigned *bms = cstart + (PRUNE ? BQ_MAXCELLS + 1 : 0);                                   // [BQ_WAVES][NR][BQ_BMW]
    float *scr = reinterpret_cast<float *>(bms + (PRUNE ? BQ_WAVES * NR * BQ_BMW : 0));       // [BQ_WAVES][6] / [BQ_WAVES]

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float *new_xyz = new_xyz_all + (size_t)b * m * 3;
    const int c_base = prm.m0 + (blockIdx.x * BQ_WAVES + wave) * CPW;
    int cnt[CPW][NR];
    int first[CPW][NR];
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            cnt[ci][r] = 0;
            first[ci][r] = 0;
        }

    bool pruned[NR];            // PRUNE: radius answered from the grid (workgroup-uniform)
#pragma unroll
    for (int r = 0; r < NR; ++r) pruned[r] = false;
    bool any_pruned = false;
    float inv_c = 0.f, glo[3] = {0.f, 0.f, 0.f};
    int G[3] = {1, 1, 1};

    for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
        const int tn = (n - t0) < BQ_TILE ? (n - t0) : BQ_TILE;
        const int tn_pad = bq_pad(tn);
        if (t0 > 0) __syncthreads();
        bq_stage_tile(xyz, t0, tn, xs, ys, zs, tid, BQ_WAVES * 64);
        __syncthreads();

        if (PRUNE) {
            // ---- which radii are answered from the grid, and the grid itself (workgroup-uniform) ----
            const float inf = __builtin_inff();
            float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
            for (int p = tid; p < tn; p += BQ_WAVES * 64) {
                const float *q = xyz + (size_t)p * 3;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    lo[a] = fminf(lo[a], q[a]);
                    hi[a] = fmaxf(hi[a], q[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
                    hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
                }
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    scr[wave * 6 + a] = lo[a];
                    scr[wave * 6 + 3 + a] = hi[a];
                }
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 3; ++a)
                for (int w = 0; w < BQ_WAVES; ++w) {
                    lo[a] = fminf(lo[a], scr[w * 6 + a]);
                    hi[a] = fmaxf(hi[a], scr[w * 6 + 3 + a]);
                }
            const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
            float rp = 0.f;                                   // largest radius answered from the grid
            if (ext > 0.f && ext < inf) {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (prm.rad[r] > 0.f && prm.rad[r] <= 0.15f * ext && prm.ns[r] > 0) {
                        pruned[r] = true;
                        rp = fmaxf(rp, prm.rad[r]);
                    }
            }
            any_pruned = rp > 0.f;
            if (any_pruned) {
                // cell edge: a hair above rp, so that |x_p - x_c| < r (as the float test sees it) puts p within one cell of c
                // along every axis whatever the roundings of the cell coordinates (margin 1e-4 against ~1e-6); enlarged
                // until the grid has at most BQ_MAXCELLS cells
                float c = rp * 1.0001f;
                for (;;) {
                    inv_c = 1.0f / c;
                    bool fits = true;
                    int cells = 1;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float f = (hi[a] - lo[a]) * inv_c;
                        if (!(f < 4096.f)) { fits = false; break; }
                        G[a] = (int)f + 1;                    // the point at hi[a] lands in cell G - 1 by the same operations
                        cells *= G[a];
                        if (cells > BQ_MAXCELLS) { fits = false; break; }
                    }
                    if (fits) break;
                    c *= 1.26f;
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) glo[a] = lo[a];
                const int cells = G[0] * G[1] * G[2];
                auto key_of = [&](float x, float y, float z) {
                    int ix = (int)((x - glo[0]) * inv_c), iy = (int)((y - glo[1]) * inv_c), iz = (int)((z - glo[2]) * inv_c);
                    ix = ix < 0 ? 0 : (ix >= G[0] ? G[0] - 1 : ix);      // (only a NaN coordinate is ever clamped)
                    iy = iy < 0 ? 0 : (iy >= G[1] ? G[1] - 1 : iy);
                    iz = iz < 0 ? 0 : (iz >= G[2] ? G[2] - 1 : iz);
                    return (iz * G[1] + iy) * G[0] + ix;
                };
                for (int e = tid; e <= cells; e += BQ_WAVES * 64) cstart[e] = 0u;
                __syncthreads();
                constexpr int PPT = BQ_PRUNE_MAXN / (BQ_WAVES * 64);
                int key[PPT];
                unsigned rank[PPT];
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    const int p = tid + i * BQ_WAVES * 64;
                    key[i] = -1;
                    rank[i] = 0u;
                    if (p < tn) {
                        const float *q = xyz + (size_t)p * 3;
                        key[i] = key_of(q[0], q[1], q[2]);
                        rank[i] = atomicAdd(&cstart[key[i]], 1u);
                    }
                }
                __syncthreads();
                // exclusive prefix over the cell counts: 4 consecutive cells per thread, wave scan, wave totals through LDS
                unsigned cnt4[4], sum4 = 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * tid + i;
                    cnt4[i] = e < cells ? cstart[e] : 0u;
                    sum4 += cnt4[i];
                }
                unsigned incl = sum4;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                unsigned *wtot = reinterpret_cast<unsigned *>(scr);
                __syncthreads();                               // (scr's bounding-box partials have been read by everyone)
                if (lane == 63) wtot[wave] = incl;
                __syncthreads();
                unsigned base = incl - sum4;
                for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * tid + i;
                    if (e < cells) cstart[e] = base;
                    base += cnt4[i];
                }
                if (tid == 0) cstart[cells] = (unsigned)tn;
                __syncthreads();
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    const int p = tid + i * BQ_WAVES * 64;
                    if (p < tn) {
                        const float *q = xyz + (size_t)p * 3;
                        srt[cstart[key[i]] + rank[i]] = make_float4(q[0], q[1], q[2], __int_as_float(p));
                    }
                }
                __syncthreads();
            }
        }

#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            const int c = c_base + ci;
            if (c >= prm.mhi) continue;
            if (PRUNE && any_pruned) {
                // ---- the grid's radii: candidates of the 27 cells, hits into bitmaps over original indices ----
                const float cx = new_xyz[(size_t)c * 3 + 0];
                const float cy = new_xyz[(size_t)c * 3 + 1];
                const float cz = new_xyz[(size_t)c * 3 + 2];
                volatile unsigned *bm = bms + (size_t)wave * NR * BQ_BMW;
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (pruned[r]) {
                        bm[r * BQ_BMW + 2 * lane] = 0u;
                        bm[r * BQ_BMW + 2 * lane + 1] = 0u;
                    }
                int ic[3];
                {
                    const float cc[3] = {cx, cy, cz};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        // a centre may lie anywhere (also outside the cloud's box, or be NaN): cell coordinate clamped to
                        // [-2, G + 1], from where the 3-cell window misses the grid
                        float f = floorf((cc[a] - glo[a]) * inv_c);
                        f = fminf(fmaxf(f, -2.f), (float)(G[a] + 1));
                        ic[a] = (int)f;
                    }
                }
                const int x0 = ic[0] - 1 < 0 ? 0 : ic[0] - 1, x1 = ic[0] + 1 >= G[0] ? G[0] - 1 : ic[0] + 1;
                if (x0 <= x1) {
                    for (int dz = -1; dz <= 1; ++dz) {
                        const int iz = ic[2] + dz;
                        if (iz < 0 || iz >= G[2]) continue;
                        for (int dy = -1; dy <= 1; ++dy) {
                            const int iy = ic[1] + dy;
                            if (iy < 0 || iy >= G[1]) continue;
                            const int row = (iz * G[1] + iy) * G[0];
                            const int s0 = (int)cstart[row + x0], s1 = (int)cstart[row + x1 + 1];
                            for (int p = s0 + lane; p < s1; p += 64) {
                                const float4 q = srt[p];
                                const float dx = cx - q.x, dy2 = cy - q.y, dz2 = cz - q.z;
                                const float d2 = (dx * dx + dy2 * dy2) + dz2 * dz2;
                                const unsigned pi = (unsigned)__float_as_int(q.w);
#pragma unroll
                                for (int r = 0; r < NR; ++r)
                                    if (pruned[r] && d2 < prm.r2[r])
                                        atomicOr(const_cast<unsigned *>(&bm[r * BQ_BMW + (pi >> 5)]), 1u << (pi & 31u));
                            }
                        }
                    }
                }
                // ---- first nsample set bits, in order; the tail padded with the first ----
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (!pruned[r]) continue;
                    unsigned w0 = bm[r * BQ_BMW + 2 * lane], w1 = bm[r * BQ_BMW + 2 * lane + 1];
                    const int pc = __popc(w0) + __popc(w1);
                    int incl = pc;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += o;
                    }
                    const int total = __builtin_amdgcn_readlane(incl, 63);
                    const unsigned long long have = __ballot(pc > 0);
                    int fill = 0;
                    if (have) {
                        const int fl = __ffsll((long long)have) - 1;
                        const int mine = w0 ? (2 * lane) * 32 + (__ffs((int)w0) - 1) : (2 * lane + 1) * 32 + (__ffs((int)w1) - 1);
                        fill = __builtin_amdgcn_readlane(mine, fl);
                    }
                    const int K = prm.ns[r];
                    int *rowp = prm.idx[r] + ((size_t)b * m + c) * K;
                    int pos = incl - pc;
                    while (w0 != 0u && pos < K) {
                        rowp[pos++] = (2 * lane) * 32 + (__ffs((int)w0) - 1);
                        w0 &= w0 - 1u;
                    }
                    while (w1 != 0u && pos < K) {
                        rowp[pos++] = (2 * lane + 1) * 32 + (__ffs((int)w1) - 1);
                        w1 &= w1 - 1u;
                    }
                    for (int s2 = (total < K ? total : K) + lane; s2 < K; s2 += 64) rowp[s2] = fill;
                    cnt[ci][r] = K;                            // closed for the scan below and for its padding
                    first[ci][r] = -1;
                }
            }
            float r2[NR];
            int ns[NR];
            int *row[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                r2[r] = prm.r2[r];
                ns[r] = prm.ns[r];
                row[r] = prm.idx[r] + ((size_t)b * m + c) * prm.ns[r];
            }
            bq_scan_centre<NR>(xs, ys, zs, tn_pad >> 8, t0, new_xyz[(size_t)c * 3 + 0], new_xyz[(size_t)c * 3 + 1], new_xyz[(size_t)c * 3 + 2],
                               r2, ns, row, cnt[ci], first[ci], lane);
        }
    }

    // pad the tail of every list with the first hit (0 when the ball is empty)
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) {
        const int c = c_base + ci;
        if (c >= prm.mhi) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (PRUNE && first[ci][r] < 0) continue;          // answered (and padded) from the grid
            bq_pad_row(prm.idx[r] + ((size_t)b * m + c) * prm.ns[r], cnt[ci][r], first[ci][r], prm.ns[r], lane);
        }
    }
}

CAPTRA_KNOB int g_bq_cpw = 0;     // experiment knob: centres per wave, 0 / 1 = one (default), 2 = two (the first form)
CAPTRA_KNOB int g_bq_prune = 0;   // opt-in: 1 = small radii of 1024..4096-point clouds from the cell grid

int launch_ball_query(int b, int n, int m, int nr, const float *radius, const int *nsample,
                      const float *new_xyz, const float *xyz, int *const *idx, const captra_launch_opts *opts, hipStream_t s) {
    if (b < 0 || n < 0 || m < 0 || nr < 1 || nr > BQ_MAXR) return -1;
    if (b == 0 || m == 0) return 0;
    BqParams prm;
    for (int r = 0; r < BQ_MAXR; ++r) {
        prm.r2[r] = 0.f;
        prm.rad[r] = 0.f;
        prm.ns[r] = 0;
        prm.idx[r] = nullptr;
    }
    for (int r = 0; r < nr; ++r) {
        if (nsample[r] < 0) return -1;
        prm.r2[r] = radius[r] * radius[r];
        prm.rad[r] = radius[r];
        prm.ns[r] = nsample[r];
        prm.idx[r] = idx[r];
    }
    const int tile_cap = bq_pad(n < BQ_TILE ? n : BQ_TILE);
    size_t shmem = (size_t)(tile_cap > 0 ? tile_cap : 256) * 3 * sizeof(float);
    const bool prune = g_bq_prune && n >= BQ_PRUNE_MINN && n <= BQ_PRUNE_MAXN && nr <= 3;   // (four bitmaps per wave do not fit in LDS)
    // One centre per wave: a wave walks its centres one after the other, so with two of them the scan's latency -- the kernel sits on
    // the serial prefix of every frame -- is paid twice.  Measured (tools/bq_ab.py, three-radius SA1 scan, us): 1 cloud 37.5 -> 20.1,
    // 16 clouds 40.9 -> 33.9, 32 clouds 70.9 -> 59.4, 64 clouds 132 -> 110, identical lists; staging the cloud once more per
    // workgroup costs less than the second walk.  The grid path (opt-in) keeps two.
    const bool one = !prune && g_bq_cpw != 2;
    const int cpw = one ? 1 : BQ_CPW;
    int wm0, wmc;
    (void)captra_centre_window(opts, m, &wm0, &wmc);
    if (wmc == 0) return 0;
    prm.m0 = wm0; prm.mhi = wm0 + wmc;
    dim3 grid((wmc + BQ_WAVES * cpw - 1) / (BQ_WAVES * cpw), b);
    dim3 block(BQ_WAVES * 64);
#define BQ_LAUNCH(NR)                                                                            \
    if (prune) {                                                                                 \
        auto kern = ball_query_kernel<NR, true>;                                                 \
        constexpr size_t extra = (size_t)(BQ_MAXCELLS + 1 + BQ_WAVES * NR * BQ_BMW) * 4 + BQ_WAVES * 6 * 4; \
        constexpr size_t cap = (size_t)BQ_PRUNE_MAXN * (12 + 16) + extra;                        \
        static CaptraDeviceOnce once;                                                            \
        if (once.first_use()) {                                                                  \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);           \
            once.done();                                                                         \
        }                                                                                        \
        CAPTRA_LAUNCH("ball_query", kern, grid, block, shmem + (size_t)tile_cap * 16 + extra, s, n, m, new_xyz, xyz, prm); \
    } else {                                                                                     \
        auto kern = ball_query_kernel<NR, false>;                                                \
        auto kern1 = ball_query_kernel<NR, false, 1>;                                            \
        static CaptraDeviceOnce once;                                                            \
        if (once.first_use()) {                                                                  \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, BQ_TILE * 12);       \
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern1),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, BQ_TILE * 12);       \
            once.done();                                                                         \
        }                                                                                        \
        if (one) { CAPTRA_LAUNCH("ball_query", kern1, grid, block, shmem, s, n, m, new_xyz, xyz, prm); } \
        else { CAPTRA_LAUNCH("ball_query", kern, grid, block, shmem, s, n, m, new_xyz, xyz, prm); }     \
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

extern "C" void captra_ball_query_set_prune(int on) { g_bq_prune = on; }
extern "C" void captra_ball_query_set_cpw(int v) { g_bq_cpw = v; }

extern "C" int captra_ball_query_ex(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                    const float *xyz, int *idx, const captra_launch_opts *opts, captra_stream_t stream) {
    int *idxs[1] = {idx};
    return launch_ball_query(b, n, m, 1, &radius, &nsample, new_xyz, xyz, idxs, opts, (hipStream_t)stream);
}
extern "C" int captra_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                 const float *xyz, int *idx, captra_stream_t stream) {
    return captra_ball_query_ex(b, n, m, radius, nsample, new_xyz, xyz, idx, nullptr, stream);
}

extern "C" int captra_ball_query_multi_ex(int b, int n, int m, int nr, const float *radius,
                                          const int *nsample, const float *new_xyz, const float *xyz,
                                          int *const *idx, const captra_launch_opts *opts, captra_stream_t stream) {
    return launch_ball_query(b, n, m, nr, radius, nsample, new_xyz, xyz, idx, opts, (hipStream_t)stream);
}
extern "C" int captra_ball_query_multi(int b, int n, int m, int nr, const float *radius,
                                       const int *nsample, const float *new_xyz, const float *xyz,
                                       int *const *idx, captra_stream_t stream) {
    return captra_ball_query_multi_ex(b, n, m, nr, radius, nsample, new_xyz, xyz, idx, nullptr, stream);
}
