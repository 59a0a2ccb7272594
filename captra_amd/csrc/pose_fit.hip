// On-device masked Procrustes pose fit for gfx950.
//
// Replaces part_fit_st_no_ransac (reference pose_utils/pose_fit.py:38-53) ->
// transform_pts_mask (pose_utils/procrustes.py:132-164) [-> transform_pts_2d_mask :213-228 ->
// rotate_pts_2d_batch :167-204] -> scale_pts_mask :117-120, translate_pts_mask :123-129, and
// rotate_pts_batch (:25-56).  The reference runs ~30 small ATen kernels per frame and moves the
// 2x2 / 3x3 cross-covariance to the HOST for torch.svd (procrustes.py:27,170), a blocking
// device->host->device round trip inside the frame loop.  Here one workgroup per
// (trajectory, part) does two reduction passes over the N points and solves the tiny problems
// in closed form on device; nothing leaves the GPU.
//
// Algebra (mask m_n = [label_n == part], c = sum m):
//   s_bar = sum m s / max(c,1), t_bar likewise                         (procrustes.py:137-138)
//   C   = sum m (t - t_bar)(s - s_bar)^T   (3x3),  Css = sum m (s - s_bar)(s - s_bar)^T
//   sym: the 2-D fit of procrustes.py:213-228 works on the (x,z) columns of s and of
//        t R; its cross-covariance is the (x,z) block of R^T C, its rotation
//        U diag(1,det(UV^T)) V^T is the rotation by atan2(M10-M01, M00+M11); R' = R * embed_y(R2)
//   scale = <R', C> / (sum m |R'(s - s_bar)|^2 + 1e-6)                  (procrustes.py:117-120)
//   trans = t_bar - scale * R' s_bar  (0 when c == 0)                   (procrustes.py:123-129)
//   valid = c > 3 and everything finite                                 (pose_fit.py:46, 26-35)
#include "common.h"

#include <math.h>

namespace {

constexpr int PF_THREADS = 256;

template <int NV>
__device__ __forceinline__ void block_reduce_sum(double (&v)[NV], double *smem /* [NV][4] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double x = v[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        v[i] = x;
    }
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[i * 4 + wave] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (smem[i * 4 + 0] + smem[i * 4 + 1]) + (smem[i * 4 + 2] + smem[i * 4 + 3]);
}

__global__ __launch_bounds__(PF_THREADS) void part_fit_st_kernel(int p, int n, int sym, int tgt_per_part,
                                                                 const int *__restrict__ labels,
                                                                 const float *__restrict__ src,
                                                                 const float *__restrict__ tgt,
                                                                 const float *__restrict__ rot,
                                                                 const float *__restrict__ given_scale,
                                                                 float *__restrict__ scale,
                                                                 float *__restrict__ trans,
                                                                 int *__restrict__ valid,
                                                                 const float *__restrict__ tgt_mean,
                                                                 const float *__restrict__ prev_scale,
                                                                 const float *__restrict__ prev_trans) {
    __shared__ double smem[15 * 4];
    const int q = blockIdx.x;
    const int bi = q / p, pi = q % p;
    const float *S = src + (size_t)q * 3 * n;
    const float *T = tgt + (size_t)(tgt_per_part ? q : bi) * 3 * n;
    // tgt_mean (B,3): the target is tgt + mean, formed here with the same single fp32 addition as the `points + points_mean`
    // tensor the reference builds (networks.py:219)
    const float tm[3] = {tgt_mean ? tgt_mean[bi * 3 + 0] : 0.f, tgt_mean ? tgt_mean[bi * 3 + 1] : 0.f, tgt_mean ? tgt_mean[bi * 3 + 2] : 0.f};
    const int *lab = labels + (size_t)bi * n;
    const int tid = threadIdx.x;

    // pass 1: count and centroids
    // (every point is loaded, members or not, and a non-member adds +0: the loads of the 16 iterations no longer wait for the
    // label they used to hide behind -- same sums bit for bit, since x + 0 = x -- and the loop pipelines)
    float fc = 0.f, fs[3] = {0.f, 0.f, 0.f}, ft[3] = {0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = tid; i < n; i += PF_THREADS) {
        const bool in = lab[i] == pi;
        fc += in ? 1.f : 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float sv = S[(size_t)a * n + i], tv = tgt_mean ? T[(size_t)a * n + i] + tm[a] : T[(size_t)a * n + i];
            fs[a] += in ? sv : 0.f;
            ft[a] += in ? tv : 0.f;
        }
    }
    double r1[7] = {fc, fs[0], fs[1], fs[2], ft[0], ft[1], ft[2]};
    block_reduce_sum<7>(r1, smem);
    const double cnt = r1[0];
    const double den = cnt > 1.0 ? cnt : 1.0;
    const double sb[3] = {r1[1] / den, r1[2] / den, r1[3] / den};
    const double tb[3] = {r1[4] / den, r1[5] / den, r1[6] / den};
    const float sbf[3] = {(float)sb[0], (float)sb[1], (float)sb[2]};
    const float tbf[3] = {(float)tb[0], (float)tb[1], (float)tb[2]};

    // pass 2: centred cross-covariance C[a][c] = sum (t_a - tb_a)(s_c - sb_c), ss = sum |s - sb|^2
    float fC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float fS[6] = {0, 0, 0, 0, 0, 0};  // sum sc sc^T: xx xy xz yy yz zz
#pragma unroll 4
    for (int i = tid; i < n; i += PF_THREADS) {
        const bool in = lab[i] == pi;
        float sc[3], tc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            sc[a] = S[(size_t)a * n + i] - sbf[a];
            tc[a] = (tgt_mean ? T[(size_t)a * n + i] + tm[a] : T[(size_t)a * n + i]) - tbf[a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = tc[a] * sc[c];
                fC[a * 3 + c] += in ? v : 0.f;
            }
        const float q0 = sc[0] * sc[0], q1 = sc[0] * sc[1], q2 = sc[0] * sc[2], q3 = sc[1] * sc[1], q4 = sc[1] * sc[2], q5 = sc[2] * sc[2];
        fS[0] += in ? q0 : 0.f; fS[1] += in ? q1 : 0.f; fS[2] += in ? q2 : 0.f;
        fS[3] += in ? q3 : 0.f; fS[4] += in ? q4 : 0.f; fS[5] += in ? q5 : 0.f;
    }
    double r2[15];
#pragma unroll
    for (int i = 0; i < 9; ++i) r2[i] = fC[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) r2[9 + i] = fS[i];
    block_reduce_sum<15>(r2, smem);

    if (tid == 0) {
        double R[9], Rf[9];
        for (int i = 0; i < 9; ++i) Rf[i] = R[i] = rot[(size_t)q * 9 + i];
        if (sym) {
            // M (2x2) = (x,z) block of R^T C
            double M[4];
            const int ax[2] = {0, 2};
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += R[k * 3 + ax[i]] * r2[k * 3 + ax[j]];
                    M[i * 2 + j] = acc;
                }
            const double a = M[0] + M[3], c = M[2] - M[1];
            const double h = sqrt(a * a + c * c);
            double cs = 1.0, sn = 0.0;
            if (h > 0.0) {
                cs = a / h;
                sn = c / h;
            } else if (h != h) {
                cs = sn = NAN;
            }
            const double R3[9] = {cs, 0, -sn, 0, 1, 0, sn, 0, cs};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double acc = 0;
                    for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * R3[k * 3 + j];
                    Rf[i * 3 + j] = acc;
                }
        }
        // numerator <R', C>; denominator sum |R'(s - sb)|^2 = <R'^T R', Css> with the actual R'
        // (the reference rotates first, then squares, so a non-orthonormal R' is honoured)
        double num = 0;
        for (int i = 0; i < 9; ++i) num += Rf[i] * r2[i];
        const double Css[9] = {r2[9], r2[10], r2[11], r2[10], r2[12], r2[13], r2[11], r2[13], r2[14]};
        double dn = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double g = 0;  // (R'^T R')_ij
                for (int k = 0; k < 3; ++k) g += Rf[k * 3 + i] * Rf[k * 3 + j];
                dn += g * Css[i * 3 + j];
            }
        const double sca = given_scale ? (double)given_scale[q] : num / (dn + 1e-6);
        double tr[3];
        for (int a = 0; a < 3; ++a) {
            const double rs = Rf[a * 3] * sb[0] + Rf[a * 3 + 1] * sb[1] + Rf[a * 3 + 2] * sb[2];
            tr[a] = cnt > 0.0 ? (tb[a] - sca * rs) : 0.0;
        }
        const float scf = (float)sca;
        const float trf[3] = {(float)tr[0], (float)tr[1], (float)tr[2]};
        double rsum = 0;
        for (int i = 0; i < 9; ++i) rsum += R[i];
        const float tsum = (trf[0] + trf[1]) + trf[2];
        const bool ok = (cnt > 3.0) && isfinite(scf) && isfinite(tsum) && isfinite(rsum);
        // prev_*: an invalid fit (<= 3 points, non-finite) keeps the previous scale / translation (networks.py:230-232)
        scale[q] = (ok || !prev_scale) ? scf : prev_scale[q];
#pragma unroll
        for (int a = 0; a < 3; ++a) trans[(size_t)q * 3 + a] = (ok || !prev_trans) ? trf[a] : prev_trans[(size_t)q * 3 + a];
        valid[q] = ok;
    }
}

// ---- 3x3 orthogonal Procrustes --------------------------------------------------------------
__device__ void jacobi_eig3(double A[9], double V[9]) {
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = A[p * 3 + q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {
                const double akp = A[k * 3 + p], akq = A[k * 3 + q];
                A[k * 3 + p] = c * akp - s * akq;
                A[k * 3 + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {
                const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                A[p * 3 + k] = c * apk - s * aqk;
                A[q * 3 + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                V[k * 3 + p] = c * vkp - s * vkq;
                V[k * 3 + q] = s * vkp + c * vkq;
            }
        }
    }
}

// R = [u1 u2 u1xu2][v1 v2 v1xv2]^T with v1,v2 the leading eigenvectors of M^T M, u_i = M v_i/|M v_i|:
// equals U diag(1,1,det(UV^T)) V^T for either sign of det(M) (see oracle/captra_oracle.c).
__device__ void kabsch3(const double M[9], double R[9]) {
    double A[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += M[k * 3 + i] * M[k * 3 + j];
            A[i * 3 + j] = acc;
        }
    jacobi_eig3(A, V);
    const double ev[3] = {A[0], A[4], A[8]};
    int o0 = 0, o1 = 1, o2 = 2;
    if (ev[o1] > ev[o0]) { int t = o0; o0 = o1; o1 = t; }
    if (ev[o2] > ev[o0]) { int t = o0; o0 = o2; o2 = t; }
    if (ev[o2] > ev[o1]) { int t = o1; o1 = o2; o2 = t; }
    double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
    for (int k = 0; k < 3; ++k) {
        v1[k] = V[k * 3 + o0];
        v2[k] = V[k * 3 + o1];
    }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    double n1 = 0, n2 = 0;
    for (int i = 0; i < 3; ++i) {
        u1[i] = M[i * 3] * v1[0] + M[i * 3 + 1] * v1[1] + M[i * 3 + 2] * v1[2];
        u2[i] = M[i * 3] * v2[0] + M[i * 3 + 1] * v2[1] + M[i * 3 + 2] * v2[2];
        n1 += u1[i] * u1[i];
        n2 += u2[i] * u2[i];
    }
    n1 = sqrt(n1);
    n2 = sqrt(n2);
    double dp = 0;
    for (int i = 0; i < 3; ++i) {
        u1[i] /= n1;
        u2[i] /= n2;
    }
    for (int i = 0; i < 3; ++i) dp += u1[i] * u2[i];
    double nn = 0;
    for (int i = 0; i < 3; ++i) {
        u2[i] -= dp * u1[i];
        nn += u2[i] * u2[i];
    }
    nn = sqrt(nn);
    for (int i = 0; i < 3; ++i) u2[i] /= nn;
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j];
}

__global__ __launch_bounds__(PF_THREADS) void procrustes_rot3_kernel(int n, const float *__restrict__ src,
                                                                     const float *__restrict__ tgt,
                                                                     float *__restrict__ rot) {
    __shared__ double smem[9 * 4];
    const int bi = blockIdx.x;
    const float *S = src + (size_t)bi * n * 3, *T = tgt + (size_t)bi * n * 3;
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += PF_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) m[a * 3 + c] += (double)T[(size_t)i * 3 + a] * (double)S[(size_t)i * 3 + c];
    }
    block_reduce_sum<9>(m, smem);
    if (threadIdx.x == 0) {
        double R[9];
        kabsch3(m, R);
        for (int i = 0; i < 9; ++i) rot[(size_t)bi * 9 + i] = (float)R[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Rotation read-out of the tracking step in one launch (one workgroup per (trajectory, part)):
//   per point:  unit y-axis (symmetric, 3 outputs)  or  ortho6d -> rotation matrix (6 outputs)
//               blocks.py:147-156 (RotationRegressor.forward), rotations.py:302-343
//   pooled   =  masked mean over the points labelled with the part; (0,1,0) / identity when none
//               networks.py:127-138
//   dR       =  from_3d(pooled) (y-axis -> frame)  or  Gram-Schmidt on the columns of pooled
//               part_dof_utils.py:137-141, rotations.py:356-387
//   R        =  R_prev * dR                          part_dof_utils.py:124-134
// Only head p on cloud (b,p) is evaluated (the reference computes all P x P and keeps the diagonal,
// networks.py:200-203).  The reference runs ~60 tiny ATen kernels for this.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize3(const float v[3], float out[3]) {  // rotations.py:302-314
    const float mag = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (mag > 1e-8f) {
        const float d = fmaxf(mag, 1e-8f);
        out[0] = v[0] / d; out[1] = v[1] / d; out[2] = v[2] / d;
    } else {
        out[0] = 1.f; out[1] = 0.f; out[2] = 0.f;
    }
}
__device__ __forceinline__ void cross3(const float u[3], const float v[3], float out[3]) {
    out[0] = u[1] * v[2] - u[2] * v[1];
    out[1] = u[2] * v[0] - u[0] * v[2];
    out[2] = u[0] * v[1] - u[1] * v[0];
}

__global__ __launch_bounds__(PF_THREADS) void rot_pool_compose_kernel(int p, int n, int sym, int diag, const float *__restrict__ raw,
                                                                      const int *__restrict__ labels,
                                                                      const float *__restrict__ prev_rot,
                                                                      float *__restrict__ rot, float *__restrict__ delta) {
    __shared__ double smem[10 * 4];
    const int q = blockIdx.x;            // = b * P + part: cloud q, head `part`
    const int bi = q / p, pi = q % p;
    const int R = sym ? 3 : 6;
    // diag: raw holds only head `part` on cloud (b, part) -- (B*P, R, N); else all P heads per cloud -- (B*P, P, R, N)
    const float *src = raw + (diag ? (size_t)q : (size_t)q * p + pi) * R * n;
    const int *lab = labels + (size_t)bi * n;
    double acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0;
    // (every point is loaded and normalised, a non-member adds +0.0 through a select -- no NaN of a degenerate non-member can
    // leak in: the loads no longer wait for the label, the loop pipelines, the sums are the same bit for bit)
#pragma unroll 4
    for (int e = threadIdx.x; e < n; e += PF_THREADS) {
        const bool in = lab[e] == pi;
        acc[9] += in ? 1.0 : 0.0;
        if (sym) {
            const float v[3] = {src[e], src[n + e], src[2 * (size_t)n + e]};
            float u[3];
            normalize3(v, u);
            acc[0] += in ? (double)u[0] : 0.0; acc[1] += in ? (double)u[1] : 0.0; acc[2] += in ? (double)u[2] : 0.0;
        } else {
            const float a[3] = {src[e], src[n + e], src[2 * (size_t)n + e]};
            const float c[3] = {src[3 * (size_t)n + e], src[4 * (size_t)n + e], src[5 * (size_t)n + e]};
            float x[3], zr[3], z[3], y[3];
            normalize3(a, x);
            cross3(x, c, zr);
            normalize3(zr, z);
            cross3(z, x, y);
            // row-major 3x3 with columns x, y, z
            acc[0] += in ? (double)x[0] : 0.0; acc[1] += in ? (double)y[0] : 0.0; acc[2] += in ? (double)z[0] : 0.0;
            acc[3] += in ? (double)x[1] : 0.0; acc[4] += in ? (double)y[1] : 0.0; acc[5] += in ? (double)z[1] : 0.0;
            acc[6] += in ? (double)x[2] : 0.0; acc[7] += in ? (double)y[2] : 0.0; acc[8] += in ? (double)z[2] : 0.0;
        }
    }
    block_reduce_sum<10>(acc, smem);
    if (threadIdx.x != 0) return;
    const float cnt = (float)acc[9];
    float dR[9];  // row-major
    if (sym) {
        float v[3];
        if (cnt > 0.f) { v[0] = (float)acc[0] / fmaxf(cnt, 1.f); v[1] = (float)acc[1] / fmaxf(cnt, 1.f); v[2] = (float)acc[2] / fmaxf(cnt, 1.f); }
        else { v[0] = 0.f; v[1] = 1.f; v[2] = 0.f; }
        float y[3], zr[3], z[3], x[3];
        const float ex[3] = {1.f, 0.f, 0.f};
        normalize3(v, y);
        cross3(ex, y, zr);
        normalize3(zr, z);
        cross3(y, z, x);
        for (int i = 0; i < 3; ++i) { dR[i * 3 + 0] = x[i]; dR[i * 3 + 1] = y[i]; dR[i * 3 + 2] = z[i]; }
    } else {
        float m[9];
        for (int i = 0; i < 9; ++i) m[i] = cnt > 0.f ? (float)acc[i] / fmaxf(cnt, 1.f) : (i % 4 == 0 ? 1.f : 0.f);
        // Gram-Schmidt on the columns (rotations.py:356-372)
        float a1[3] = {m[0], m[3], m[6]}, a2[3] = {m[1], m[4], m[7]}, a3[3] = {m[2], m[5], m[8]};
        float u2[3], u3[3];
        auto dot = [](const float *u, const float *v) { return (u[0] * v[0] + u[1] * v[1]) + u[2] * v[2]; };
        const float k12 = dot(a1, a2) / fmaxf(dot(a1, a1), 1e-8f);
        for (int i = 0; i < 3; ++i) u2[i] = a2[i] - k12 * a1[i];
        const float k13 = dot(a1, a3) / fmaxf(dot(a1, a1), 1e-8f);
        const float k23 = dot(u2, a3) / fmaxf(dot(u2, u2), 1e-8f);
        for (int i = 0; i < 3; ++i) u3[i] = (a3[i] - k13 * a1[i]) - k23 * u2[i];
        float c1[3], c2[3], c3[3];
        normalize3(a1, c1); normalize3(u2, c2); normalize3(u3, c3);
        for (int i = 0; i < 3; ++i) { dR[i * 3 + 0] = c1[i]; dR[i * 3 + 1] = c2[i]; dR[i * 3 + 2] = c3[i]; }
    }
    const float *Rp = prev_rot + (size_t)q * 9;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float v = 0.f;
            for (int k = 0; k < 3; ++k) v += Rp[i * 3 + k] * dR[k * 3 + j];
            rot[(size_t)q * 9 + i * 3 + j] = v;
        }
    if (delta != nullptr)
        for (int i = 0; i < 9; ++i) delta[(size_t)q * 9 + i] = dR[i];
}

}  // namespace

extern "C" int captra_rot_pool_compose(int b, int p, int n, int sym, int diag_only, const float *raw, const int *labels,
                                       const float *prev_rot, float *rot, float *delta, captra_stream_t stream) {
    if (b < 0 || p < 1 || n < 0) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("rot_pool_compose", rot_pool_compose_kernel, dim3(b * p), dim3(PF_THREADS), 0, (hipStream_t)stream, p, n, sym,
                  diag_only, raw, labels, prev_rot, rot, delta);
    return captra_last_error();
}

extern "C" int captra_part_fit_st(int b, int p, int n, int sym, const int *labels, const float *src,
                                  const float *tgt, int tgt_per_part, const float *rot, const float *given_scale,
                                  float *scale, float *trans, int *valid, captra_stream_t stream) {
    if (b < 0 || p < 1 || n < 0) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("part_fit_st", part_fit_st_kernel, dim3(b * p), dim3(PF_THREADS), 0, (hipStream_t)stream, p, n,
                  sym, tgt_per_part, labels, src, tgt, rot, given_scale, scale, trans, valid, (const float *)nullptr,
                  (const float *)nullptr, (const float *)nullptr);
    return captra_last_error();
}

// The track loop's form of captra_part_fit_st (networks.py:219-232 in one launch): the target is pts (B,3,N) + pts_mean (B,3)
// -- formed in the kernel, no (B,3,N) temporary -- and a part whose fit is invalid keeps prev_scale (B,P) / prev_trans (B,P,3).
extern "C" int captra_part_fit_st_track(int b, int p, int n, int sym, const int *labels, const float *src, const float *pts,
                                        const float *pts_mean, const float *rot, const float *prev_scale, const float *prev_trans,
                                        float *scale, float *trans, int *valid, captra_stream_t stream) {
    if (b < 0 || p < 1 || n < 0) return -1;
    if (b == 0) return 0;
    CAPTRA_LAUNCH("part_fit_st", part_fit_st_kernel, dim3(b * p), dim3(PF_THREADS), 0, (hipStream_t)stream, p, n,
                  sym, 0, labels, src, pts, rot, (const float *)nullptr, scale, trans, valid, pts_mean, prev_scale, prev_trans);
    return captra_last_error();
}

extern "C" int captra_procrustes_rot3(int nb, int n, const float *src, const float *tgt, float *rot,
                                      captra_stream_t stream) {
    if (nb < 0 || n < 0) return -1;
    if (nb == 0) return 0;
    CAPTRA_LAUNCH("procrustes_rot3", procrustes_rot3_kernel, dim3(nb), dim3(PF_THREADS), 0, (hipStream_t)stream, n,
                  src, tgt, rot);
    return captra_last_error();
}
