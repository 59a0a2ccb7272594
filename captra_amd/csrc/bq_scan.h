// The ball query's index-order scan of one centre by one wave (reference ball_query_gpu.cu:9-45: first nsample points with
// d2 < r^2 in index order, the tail padded with the first hit), shared by ball_query_kernel (csrc/ball_query.hip) and the level-1
// stream kernel (csrc/sa_bf16.hip).  The cloud tile sits in LDS as three coordinate planes in which the 64-point chunks are
// interleaved four by four (element ((chunk / 4) * 64 + lane) * 4 + chunk % 4): one ds_read_b128 per plane feeds 256 distance
// tests, computed with packed fp32 instructions, unfused: ((cx-x)^2 + (cy-y)^2) + (cz-z)^2, strict '<'.  Slots past the end of the
// cloud hold +inf and never hit.
#pragma once
#include "common.h"

typedef float bq_f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int bq_pad(int v) { return (v + 255) & ~255; }

// stage points [t0, t0 + tn) of one cloud (point-major xyz) into the three planes; nthreads threads cooperate
__device__ __forceinline__ void bq_stage_tile(const float *xyz, int t0, int tn, float *xs, float *ys, float *zs, int tid, int nthreads) {
    const int tn_pad = bq_pad(tn);
    for (int p = tid; p < tn_pad; p += nthreads) {
        const bool inb = p < tn;
        const float *q = xyz + (size_t)(t0 + (inb ? p : 0)) * 3;
        const float inf = __builtin_inff();
        const int chunk = p >> 6;
        const int a = (((chunk >> 2) << 6) + (p & 63)) * 4 + (chunk & 3);
        xs[a] = inb ? q[0] : inf;
        ys[a] = inb ? q[1] : inf;
        zs[a] = inb ? q[2] : inf;
    }
}

// One wave, one centre, one staged tile: hits of radius r land in row[r][cnt[r] + rank among this chunk's hits] (ordered
// compaction: the compare's own 64-bit mask + v_mbcnt), cnt[r] counts every hit seen, first[r] = the first hit's index.  Radii whose
// list is full are skipped; the scan stops when all are (wave-uniform early exit).  `groups` = 256-point groups in the tile, `t0` =
// the tile's first point.
template <int NR>
__device__ __forceinline__ void bq_scan_centre(const float *xs, const float *ys, const float *zs, int groups, int t0, float cx, float cy,
                                               float cz, const float (&r2)[NR], const int (&ns)[NR], int *const (&row)[NR],
                                               int (&cnt)[NR], int (&first)[NR], int lane) {
    bool open = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) open = open || (cnt[r] < ns[r]);
    if (!open) return;
    const bq_f32x4 cx4 = {cx, cx, cx, cx}, cy4 = {cy, cy, cy, cy}, cz4 = {cz, cz, cz, cz};
    for (int g = 0; g < groups && open; ++g) {
        const bq_f32x4 dx = cx4 - reinterpret_cast<const bq_f32x4 *>(xs)[g * 64 + lane];
        const bq_f32x4 dy = cy4 - reinterpret_cast<const bq_f32x4 *>(ys)[g * 64 + lane];
        const bq_f32x4 dz = cz4 - reinterpret_cast<const bq_f32x4 *>(zs)[g * 64 + lane];
        const bq_f32x4 d2 = (dx * dx + dy * dy) + dz * dz;  // unfused: the build runs with -ffp-contract=off
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (!open) break;
            const int k0 = t0 + g * 256 + h * 64;
            bool any_open = false;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (cnt[r] < ns[r]) {
                    const bool hit = d2[h] < r2[r];
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        // rank of this lane among the hits = hits in lower lanes (v_mbcnt), written through a
                        // uniform row pointer + unsigned 32-bit slot (scalar base + VGPR offset addressing)
                        const unsigned pos = (unsigned)cnt[r] +
                                             __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (cnt[r] == 0) first[r] = k0 + (__ffsll((long long)mask) - 1);
                        if (hit && pos < (unsigned)ns[r]) row[r][pos] = k0 + lane;
                        cnt[r] += __popcll(mask);
                    }
                    any_open = any_open || (cnt[r] < ns[r]);
                }
            }
            open = any_open;
        }
    }
}

// One wave, NC centres x NR radii, one staged tile: every 256-point group is read once (three ds_read_b128) and tested against all
// NC centres -- NC independent chains of packed arithmetic behind one LDS latency -- then each (centre, radius) list takes its four
// chunks' hits in index order exactly as in bq_scan_centre.  For a wave that owns several centres and shares its SIMD with few
// other waves (the level-1 stream kernel's consumers: two waves per SIMD), where one centre at a time waits out every LDS read.
// List (c, r) = rows[r] + c * ns[r] (consecutive centres: consecutive rows); cnt = ns closes a slot from the start.  Radii ascending
// (r2[NR - 1] the largest: the quick reject tests against it).
template <int NC, int NR>
__device__ __forceinline__ void bq_scan_centres(const float *xs, const float *ys, const float *zs, int groups, int t0, const float (&cx)[NC],
                                                const float (&cy)[NC], const float (&cz)[NC], const float (&r2)[NR], const int (&ns)[NR],
                                                int *const (&rows)[NR], int (&cnt)[NC][NR], int (&first)[NC][NR], int lane) {
    bool any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < NR; ++r) any = any || (cnt[c][r] < ns[r]);
    for (int g = 0; g < groups && any; ++g) {
        const bq_f32x4 X = reinterpret_cast<const bq_f32x4 *>(xs)[g * 64 + lane];
        const bq_f32x4 Y = reinterpret_cast<const bq_f32x4 *>(ys)[g * 64 + lane];
        const bq_f32x4 Z = reinterpret_cast<const bq_f32x4 *>(zs)[g * 64 + lane];
        bq_f32x4 d2[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bq_f32x4 dx = cx[c] - X, dy = cy[c] - Y, dz = cz[c] - Z;
            d2[c] = (dx * dx + dy * dy) + dz * dz;      // unfused: the build runs with -ffp-contract=off
        }
        any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // quick reject: none of the centre's 256 tests of this group is inside the largest radius (the small radii of a level
            // hit in a handful of the cloud's 64-point chunks: one min3 pair + one compare instead of four compare / branch pairs)
            const float dmin4 = fminf(fminf(d2[c][0], d2[c][1]), fminf(d2[c][2], d2[c][3]));
            if (__ballot(dmin4 < r2[NR - 1]) == 0ull) {
#pragma unroll
                for (int r = 0; r < NR; ++r) any = any || (cnt[c][r] < ns[r]);
                continue;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (cnt[c][r] < ns[r]) {
                        const bool hit = d2[c][h] < r2[r];
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            const int k0 = t0 + g * 256 + h * 64;
                            const unsigned pos = (unsigned)cnt[c][r] +
                                                 __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            if (cnt[c][r] == 0) first[c][r] = k0 + (__ffsll((long long)mask) - 1);
                            if (hit && pos < (unsigned)ns[r]) (rows[r] + (size_t)c * ns[r])[pos] = k0 + lane;
                            cnt[c][r] += __popcll(mask);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) any = any || (cnt[c][r] < ns[r]);
        }
    }
}

// pad the tail of a list with the first hit (0 when the ball is empty)
__device__ __forceinline__ void bq_pad_row(int *row, int cnt, int first, int ns, int lane) {
    const int have = cnt < ns ? cnt : ns;
    for (int s = have + lane; s < ns; s += 64) row[s] = first;
}
