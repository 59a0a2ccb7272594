// Ball crop of a depth frame around a predicted centre, for gfx950 — the candidate extraction of the on-the-fly re-crop
// of NOCS tracking (reference datasets/nocs_data/nocs_data_process.py:92-109, 151-163 `crop_ball_from_depth_image` /
// `crop_ball_from_pts`; nocs_utils.py:5-33 `backproject`): inside the image-space box of the ball, back-project every
// pixel with a valid depth, keep those within `radius` of the centre, IN ROW-MAJOR PIXEL ORDER (the order decides which
// point furthest-point sampling starts from).  The reference does this in numpy on the host per tracked instance; the
// torch restatement in captra_amd/nocs_otf.py costs ~0.35 ms and two host syncs per instance.  Here one workgroup per
// instance walks its box 1024 pixels at a time and compacts the members in order (wave ballot + LDS wave offsets).
//
// Arithmetic is float64 like numpy's, written out operation by operation (no FMA contraction: the build runs with
// -ffp-contract=off):  ray = Kinv (col, H - row, 1);  p = ray * z / ray_z;  p = (p_x, p_y, -p_z) * 0.001;
// member  <=>  sqrt(((p - c)_x^2 + (p - c)_y^2) + (p - c)_z^2) <= radius.
#include "common.h"

#include <math.h>

namespace {

constexpr int CB_T = 1024;

__global__ __launch_bounds__(CB_T) void crop_ball_kernel(int h, int w, int cap, const int *__restrict__ depth_all,
                                                         const unsigned char *__restrict__ mask_all,
                                                         const int *__restrict__ box_all, const double *__restrict__ center_all,
                                                         const double *__restrict__ radius_all, const double *__restrict__ kinv,
                                                         double *__restrict__ pts_all, unsigned char *__restrict__ obj_all,
                                                         int *__restrict__ pix_all, int *__restrict__ counts) {
    __shared__ int wave_tot[2][16];
    const int b = blockIdx.x;
    const int *depth = depth_all + (size_t)b * h * w;
    const unsigned char *mask = mask_all + (size_t)b * h * w;
    const int r0 = box_all[b * 4 + 0], c0 = box_all[b * 4 + 1], r1 = box_all[b * 4 + 2], c1 = box_all[b * 4 + 3];
    const double cx = center_all[b * 3 + 0], cy = center_all[b * 3 + 1], cz = center_all[b * 3 + 2];
    const double rad = radius_all[b];
    double k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = kinv[i];
    double *pts = pts_all + (size_t)b * cap * 3;
    unsigned char *obj = obj_all + (size_t)b * cap;
    int *pix = pix_all + (size_t)b * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
    const long long total = (bw > 0 && bh > 0) ? (long long)bw * bh : 0;
    int base = 0, valid = 0, it = 0;
    for (long long t0 = 0; t0 < total; t0 += CB_T, ++it) {
        const long long t = t0 + tid;
        bool member = false, has_depth = false;
        double px = 0., py = 0., pz = 0.;
        int row = 0, col = 0;
        if (t < total) {
            row = r0 + (int)(t / bw);
            col = c0 + (int)(t % bw);
            const int d = depth[(size_t)row * w + col];
            if (d > 0) {
                has_depth = true;
                const double u = (double)col, v = (double)(h - row);
                const double rx = (k[0] * u + k[1] * v) + k[2];
                const double ry = (k[3] * u + k[4] * v) + k[5];
                const double rz = (k[6] * u + k[7] * v) + k[8];
                const double z = (double)(float)d;                 // depth[idxs].astype(np.float32)
                px = (rx * z / rz) * 0.001;
                py = (ry * z / rz) * 0.001;
                pz = -(rz * z / rz) * 0.001;
                const double dx = px - cx, dy = py - cy, dz = pz - cz;
                member = sqrt((dx * dx + dy * dy) + dz * dz) <= rad;
            }
        }
        const unsigned long long m = __ballot(member), v = __ballot(has_depth);
        if (lane == 0) wave_tot[it & 1][wave] = __popcll(m) | (__popcll(v) << 8);
        __syncthreads();
        int before = 0, all = 0, allv = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = wave_tot[it & 1][i];
            before += i < wave ? (e & 0xFF) : 0;
            all += e & 0xFF;
            allv += e >> 8;
        }
        if (member) {
            const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) {
                pts[(size_t)pos * 3 + 0] = px; pts[(size_t)pos * 3 + 1] = py; pts[(size_t)pos * 3 + 2] = pz;
                obj[pos] = mask[(size_t)row * w + col];
                pix[pos] = row * w + col;
            }
        }
        base += all;
        valid += allv;
    }
    if (tid == 0) { counts[b * 2 + 0] = base; counts[b * 2 + 1] = valid; }
}

// The crop's image-space box from the device-resident pose (reference nocs_data_process.py:136-148 `proj_corners`; model.py:425-452:
// the ball is centred on the pose predicted for the previous frame, radius = radius_factor x its scale): float64, operation by
// operation as numpy evaluates captra_amd/nocs_otf.py::proj_corners_batch -- r = max(radius_factor * scale, 0.05); the 8 corners of
// c +- r times 1000; (-x / z, -y / z, 1); u = (K00 x + K01 y) + K02, v = (K10 x + K11 y) + K12 truncated to int32; rows = h - v, cols = u;
// min / max over the corners, clamped to the image.  One thread per instance: the box, centre and radius crop_ball_kernel reads.
__global__ void crop_box_kernel(int b, int h, int w, double radius_factor, const float *__restrict__ trans, const float *__restrict__ scale,
                                const double *__restrict__ kmat, int *__restrict__ box, double *__restrict__ center,
                                double *__restrict__ radius) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    const double c[3] = {(double)trans[i * 3 + 0], (double)trans[i * 3 + 1], (double)trans[i * 3 + 2]};
    double r = radius_factor * (double)scale[i];
    r = r > 0.05 ? r : 0.05;                     // numpy.maximum(r, 0.05) (NaN propagates there; a NaN pose has no crop either way)
    int rmin = 0x7fffffff, rmax = -0x7fffffff - 1, cmin = 0x7fffffff, cmax = -0x7fffffff - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double bx = ((e & 1) ? c[0] + r : c[0] - r) * 1000.0;
        const double by = ((e & 2) ? c[1] + r : c[1] - r) * 1000.0;
        const double bz = ((e & 4) ? c[2] + r : c[2] - r) * 1000.0;
        const double hx = (-bx) / bz, hy = (-by) / bz, hz = -((-bz) / bz);
        const double u = (kmat[0] * hx + kmat[1] * hy) + kmat[2] * hz;
        const double v = (kmat[3] * hx + kmat[4] * hy) + kmat[5] * hz;
        const int row = h - (int)v, col = (int)u;
        rmin = row < rmin ? row : rmin; rmax = row > rmax ? row : rmax;
        cmin = col < cmin ? col : cmin; cmax = col > cmax ? col : cmax;
    }
    box[i * 4 + 0] = rmin > 0 ? rmin : 0;
    box[i * 4 + 1] = cmin > 0 ? cmin : 0;
    box[i * 4 + 2] = rmax < h - 1 ? rmax : h - 1;
    box[i * 4 + 3] = cmax < w - 1 ? cmax : w - 1;
    center[i * 3 + 0] = c[0]; center[i * 3 + 1] = c[1]; center[i * 3 + 2] = c[2];
    radius[i] = r;
}

// ---- the re-crop's candidate lists and its read-out without the host --------------------------------------
// What the host did between the crop and the sampler, from the member counts it fetched (captra_amd/nocs_otf.py, reference
// data_utils.py:138-162 farthest_point_sample's caller / nocs_data_process.py:92-109): the candidate list of an instance is its
// member table repeated until it holds at least num_points entries (list doubling: candidate j = member j mod count, length =
// count * 2^k), thinned by a random permutation when longer than 5 num_points (the host's generator: a RARE path here).
// otf_candidates_kernel builds the lists (fp32 coordinates, the sampler's input) and their lengths on the device and raises the
// rare-path word; otf_finish_kernel turns the sampler's picks into the frame's tensors (reference nocs_data_process.py:43-50,
// 227-236: points, labels 0 = object / 1 = background, ground-truth NOCS of the object's points) in the layouts the networks read.
__device__ __forceinline__ int otf_list_length(int c, int num_points) {
    int len = c;
    while (len < num_points) len *= 2;
    return len;
}

__global__ __launch_bounds__(256) void otf_candidates_kernel(int cap, int stride, int num_points, const double *__restrict__ pts,
                                                             const int *__restrict__ counts, float *__restrict__ cand,
                                                             int *__restrict__ lens, int *__restrict__ info) {
    const int b = blockIdx.y;
    const int c = counts[b * 2];
    const int cc = c < 1 ? 1 : (c > stride ? stride : c);           // rare rows clamped: every access stays inside its buffer
    const int len = otf_list_length(cc, num_points);
    const int keep = len < stride ? len : stride;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        lens[b] = keep;
        if (c < 10 || c > stride || len > stride) atomicOr(info, 1);
        atomicMax(info + 1, c > cap ? cap : otf_list_length(c < 1 ? 1 : c, num_points));
    }
    const double *pb = pts + (size_t)b * cap * 3;
    float *cb = cand + (size_t)b * stride * 3;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < stride; j += gridDim.x * 256) {
        float x = 0.f, y = 0.f, z = 0.f;
        if (j < keep) {
            const double *q = pb + (size_t)(j % cc) * 3;
            x = (float)q[0]; y = (float)q[1]; z = (float)q[2];
        }
        cb[(size_t)j * 3 + 0] = x; cb[(size_t)j * 3 + 1] = y; cb[(size_t)j * 3 + 2] = z;
    }
}

__global__ __launch_bounds__(256) void otf_finish_kernel(int cap, int stride, int n, const double *__restrict__ pts,
                                                         const unsigned char *__restrict__ obj, const int *__restrict__ counts,
                                                         const int *__restrict__ picks, const float *__restrict__ mean,
                                                         const double *__restrict__ rot, const double *__restrict__ trans,
                                                         const double *__restrict__ scale, float *__restrict__ points_cn,
                                                         long long *__restrict__ labels, float *__restrict__ nocs_cn) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = counts[b * 2];
    const int cc = c < 1 ? 1 : (c > stride ? stride : c);
    const int m = picks[(size_t)b * n + i] % cc;                      // candidate -> member
    const double *q = pts + ((size_t)b * cap + m) * 3;
    const bool o = obj[(size_t)b * cap + m] != 0;
    const double *R = rot + (size_t)b * 9, *t = trans + (size_t)b * 3;
    const double s = scale[b];
    const double x0 = (q[0] - t[0]) / s, x1 = (q[1] - t[1]) / s, x2 = (q[2] - t[2]) / s;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        points_cn[((size_t)b * 3 + a) * n + i] = (float)q[a] - mean[b * 3 + a];
        nocs_cn[((size_t)b * 3 + a) * n + i] = o ? (float)((x0 * R[a] + x1 * R[3 + a]) + x2 * R[6 + a]) : 0.f;
    }
    labels[(size_t)b * n + i] = o ? 0 : 1;
}

}  // namespace

extern "C" int captra_crop_box(int b, int h, int w, double radius_factor, const float *trans, const float *scale, const double *kmat,
                               int *box, double *center, double *radius, captra_stream_t stream) {
    if (b < 0 || h < 1 || w < 1) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("crop_ball", crop_box_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, b, h, w, radius_factor, trans,
                  scale, kmat, box, center, radius);
    return captra_last_error();
}

extern "C" int captra_crop_ball(int b, int h, int w, int cap, const int *depth, const unsigned char *mask, const int *box,
                                const double *center, const double *radius, const double *kinv, double *pts,
                                unsigned char *obj, int *pix, int *counts, captra_stream_t stream) {
    if (b < 0 || h < 1 || w < 1 || cap < 1) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("crop_ball", crop_ball_kernel, dim3(b), dim3(CB_T), 0, (hipStream_t)stream, h, w, cap, depth, mask, box,
                  center, radius, kinv, pts, obj, pix, counts);
    return captra_last_error();
}

// The candidate lists of a re-crop and the ragged sampler's per-cloud counts, from the crop's device-resident member counts (see the
// kernels above).  cand (B,stride,3) fp32, lens (B,), info: 4 ints, zeroed here on the stream, [0] = a rare-path instance was met
// (< 10 members, or a list longer than `stride`), [1] = the longest list.  num_points <= stride.
extern "C" int captra_otf_candidates(int b, int cap, int stride, int num_points, const double *pts, const int *counts, float *cand,
                                     int *lens, int *info, captra_stream_t stream) {
    if (b < 0 || cap < 1 || stride < 1 || num_points < 1 || num_points > stride) return -1;
    if (b == 0) return 0;
    if (captra_zero_async(info, 16, (hipStream_t)stream) != 0) return -1;
    CAPTRA_LAUNCH("crop_ball", otf_candidates_kernel, dim3((unsigned)((stride + 1023) / 1024), b), dim3(256), 0, (hipStream_t)stream, cap, stride,
                  num_points, pts, counts, cand, lens, info);
    return captra_last_error();
}

// picks (B,n) = the sampler's indices into the candidate lists -> points_cn (B,3,n) fp32 = member coordinates - mean (B,3),
// labels (B,n) int64 (0 object / 1 background), nocs_cn (B,3,n) fp32 = ((p - trans) / scale) R of the object's points in float64, 0 elsewhere.
extern "C" int captra_otf_finish(int b, int cap, int stride, int n, const double *pts, const unsigned char *obj, const int *counts,
                                 const int *picks, const float *mean, const double *rot, const double *trans, const double *scale,
                                 float *points_cn, long long *labels, float *nocs_cn, captra_stream_t stream) {
    if (b < 0 || cap < 1 || stride < 1 || n < 1) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("crop_ball", otf_finish_kernel, dim3((unsigned)((n + 255) / 256), b), dim3(256), 0, (hipStream_t)stream, cap, stride, n, pts, obj,
                  counts, picks, mean, rot, trans, scale, points_cn, labels, nocs_cn);
    return captra_last_error();
}
