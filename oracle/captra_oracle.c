/*
 * captra_oracle.c — CPU restatement of the reference's algorithms for the CAPTRA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (captra_amd/) may import, link or call
 * this file; it exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * have an independent checker that travels to the GPU box (the reference's Python does not).
 *
 * Every function restates, in plain C with the same operation order, the reference code cited
 * above it (paths relative to the reference checkout).  Pinning: tests/test_oracle_golden.py
 * checks these functions against tests/golden/ *.npz, which were produced by importing the
 * reference's own CPU path (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp (oracle/Makefile).  -ffp-contract=off keeps
 * every fp32 operation separately rounded (the contract of SURVEY.md §2.2); -mfma only makes
 * the EXPLICIT fmaf() calls of the MLP restatement single instructions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

static inline float d2f(float ax, float ay, float az, float bx, float by, float bz) {
    /* (a-b)^2 summed as ((dx*dx + dy*dy) + dz*dz): sampling_gpu.cu:133, ball_query_gpu.cu:33,
     * interpolate_gpu.cu:107 (all the same expression shape) */
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

/* ---- furthest point sampling --------------------------------------------------------------
 * sampling_gpu.cu:93-209 with the tie rule of the importable CPU path (pointnet_utils.py:137,
 * torch.max -> first index): start at index 0, temp[k] = min(d, temp[k]), argmax with strict
 * '>' over ascending k. */
EXPORT void oracle_fps(int b, int n, int m, const float *xyz_all, float *temp_all, int *idx_all) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < b; ++bi) {
        const float *xyz = xyz_all + (size_t)bi * n * 3;
        float *temp = temp_all + (size_t)bi * n;
        int *idx = idx_all + (size_t)bi * m;
        if (m <= 0) continue;
        int old = 0;
        idx[0] = 0;
        for (int j = 1; j < m; ++j) {
            int besti = 0;
            float best = -1.0f;
            float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
            for (int k = 0; k < n; ++k) {
                float d = d2f(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], x1, y1, z1);
                float d2 = d < temp[k] ? d : temp[k]; /* min(d, temp[k]) */
                temp[k] = d2;
                if (d2 > best) {
                    best = d2;
                    besti = k;
                }
            }
            old = besti;
            idx[j] = old;
        }
    }
}

/* ---- ball query: ball_query_gpu.cu:9-45 ---------------------------------------------------- */
EXPORT void oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz_all,
                              const float *xyz_all, int *idx_all) {
    const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int pt = 0; pt < m; ++pt) {
            const float *c = new_xyz_all + ((size_t)bi * m + pt) * 3;
            const float *xyz = xyz_all + (size_t)bi * n * 3;
            int *idx = idx_all + ((size_t)bi * m + pt) * nsample;
            for (int l = 0; l < nsample; ++l) idx[l] = 0; /* caller pre-zeroes: pointnet2_utils.py:261 */
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                float d2 = d2f(c[0], c[1], c[2], xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) idx[l] = k;
                    idx[cnt] = k;
                    ++cnt;
                }
            }
        }
    }
}

/* ---- group / gather: group_points_gpu.cu:47-66, sampling_gpu.cu:8-24 ----------------------- */
EXPORT void oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                const int *idx, float *out) {
    const size_t npos = (size_t)npoints * nsample;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (size_t p = 0; p < npos; ++p)
                out[((size_t)bi * c + ci) * npos + p] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * npos + p]];
}

/* group_points_gpu.cu:8-25 (atomicAdd scatter; serial order here) */
EXPORT void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                     const int *idx, float *grad_points) {
    const size_t npos = (size_t)npoints * nsample;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (size_t p = 0; p < npos; ++p)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * npos + p]] +=
                    grad_out[((size_t)bi * c + ci) * npos + p];
}

EXPORT void oracle_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                                 float *out) {
    oracle_group_points(b, c, n, npoints, 1, points, idx, out);
}

/* sampling_gpu.cu:46-63 */
EXPORT void oracle_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                      float *grad_points) {
    oracle_group_points_grad(b, c, n, npoints, 1, grad_out, idx, grad_points);
}

/* ---- three_nn: interpolate_gpu.cu:81-124 (double comparators, squared distances out) -------- */
EXPORT void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                            int *idx) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int pt = 0; pt < n; ++pt) {
            const float *u = unknown + ((size_t)bi * n + pt) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                float d = d2f(u[0], u[1], u[2], kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *dd = dist2 + ((size_t)bi * n + pt) * 3;
            int *ii = idx + ((size_t)bi * n + pt) * 3;
            dd[0] = (float)best1; dd[1] = (float)best2; dd[2] = (float)best3;
            ii[0] = besti1; ii[1] = besti2; ii[2] = besti3;
        }
    }
}

/* ---- knn: interpolate_gpu.cu:9-57 ----------------------------------------------------------- */
EXPORT int oracle_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
                      int *idx) {
    if (k < 1 || k > 200) return -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int pt = 0; pt < n; ++pt) {
            const float *u = unknown + ((size_t)bi * n + pt) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            double best[200];
            int besti[200];
            for (int i = 0; i < k; ++i) {
                best[i] = 1e40;
                besti[i] = 0;
            }
            for (int i = 0; i < m; ++i) {
                float d = d2f(u[0], u[1], u[2], kn[i * 3 + 0], kn[i * 3 + 1], kn[i * 3 + 2]);
                for (int j = 0; j < k; ++j) {
                    if (d < best[j]) {
                        for (int l = k - 1; l > j; --l) {
                            best[l] = best[l - 1];
                            besti[l] = besti[l - 1];
                        }
                        best[j] = d;
                        besti[j] = i;
                        break;
                    }
                }
            }
            for (int i = 0; i < k; ++i) {
                idx[((size_t)bi * n + pt) * k + i] = besti[i];
                dist2[((size_t)bi * n + pt) * k + i] = (float)best[i];
            }
        }
    }
    return 0;
}

/* ---- three_interpolate: interpolate_gpu.cu:149-169 ------------------------------------------ */
EXPORT void oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                     const float *weight, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *row = points + ((size_t)bi * c + ci) * m;
            for (int p = 0; p < n; ++p) {
                const int *id = idx + ((size_t)bi * n + p) * 3;
                const float *w = weight + ((size_t)bi * n + p) * 3;
                float a = w[0] * row[id[0]], bb = w[1] * row[id[1]], cc = w[2] * row[id[2]];
                out[((size_t)bi * c + ci) * n + p] = (a + bb) + cc;
            }
        }
}

/* interpolate_gpu.cu:192-214 */
EXPORT void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *row = grad_points + ((size_t)bi * c + ci) * m;
            for (int p = 0; p < n; ++p) {
                const int *id = idx + ((size_t)bi * n + p) * 3;
                const float *w = weight + ((size_t)bi * n + p) * 3;
                float g = grad_out[((size_t)bi * c + ci) * n + p];
                row[id[0]] += g * w[0];
                row[id[1]] += g * w[1];
                row[id[2]] += g * w[2];
            }
        }
}

/* ---- canonicalisation: networks.py:38-41 / 184-187 ------------------------------------------
 * cam = pts + mean; cam = cam - t; cam = R^T cam; cam = cam / s.  The 3-term dot product is
 * summed left to right, each product and sum rounded (the HIP kernel uses the same order; the
 * reference's torch.matmul may order/fuse differently — a <= 1 ulp effect covered by the golden
 * tolerance). */
EXPORT void oracle_canonicalize(int b, int p, int n, const float *pts, const float *mean, const float *rot,
                                const float *trans, const float *scale, float *out_cn, float *out_n3) {
#pragma omp parallel for schedule(static)
    for (int q = 0; q < b * p; ++q) {
        const int bi = q / p;
        const float *R = rot + (size_t)q * 9;
        const float *t = trans + (size_t)q * 3;
        const float s = scale[q];
        for (int i = 0; i < n; ++i) {
            float v[3];
            for (int a = 0; a < 3; ++a) v[a] = (pts[((size_t)bi * 3 + a) * n + i] + mean[bi * 3 + a]) - t[a];
            for (int a = 0; a < 3; ++a) {
                /* (R^T v)[a] = sum_j R[j][a] v[j] */
                float acc = (R[0 * 3 + a] * v[0] + R[1 * 3 + a] * v[1]) + R[2 * 3 + a] * v[2];
                float o = acc / s;
                if (out_cn) out_cn[((size_t)q * 3 + a) * n + i] = o;
                if (out_n3) out_n3[((size_t)q * n + i) * 3 + a] = o;
            }
        }
    }
}

/* ---- shared MLP layer: Conv 1x1 + folded BatchNorm + activation -----------------------------
 * pointnet_utils.py:242-245 / 296-298; arithmetic contract of include/captra_hip.h:
 * acc = bias; acc = fmaf(W[co][k], x[k][l], acc) for k ascending; then the activation. */
static inline float act_apply(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return 1.0f / (1.0f + expf(-v)) - 0.5f;
    return v;
}

EXPORT void oracle_pointwise_mlp(int b, int cin, int cout, long long l, const float *x, const float *wt,
                                 const float *bias, int act, float *y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int co = 0; co < cout; ++co) {
            float *yr = y + ((size_t)bi * cout + co) * l;
            for (long long p = 0; p < l; ++p) yr[p] = bias[co];
            for (int k = 0; k < cin; ++k) {
                const float w = wt[(size_t)k * cout + co];
                const float *xr = x + ((size_t)bi * cin + k) * l;
                for (long long p = 0; p < l; ++p) yr[p] = fmaf(w, xr[p], yr[p]);
            }
            for (long long p = 0; p < l; ++p) yr[p] = act_apply(yr[p], act);
        }
}

/* group + "-= centre" + concat [feat, xyz] (pointnet_utils.py:234-240) feeding the first layer */
EXPORT void oracle_sa_group(int b, int n, int m, int k, int cfeat, const float *feat, const float *xyz_cn,
                            const float *new_xyz, const int *idx, float *x) {
    const int cin = cfeat + 3;
    const size_t mk = (size_t)m * k;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < cin; ++ci)
            for (size_t p = 0; p < mk; ++p) {
                const int id = idx[(size_t)bi * mk + p];
                float v;
                if (ci < cfeat) {
                    v = feat[((size_t)bi * cfeat + ci) * n + id];
                } else {
                    const int a = ci - cfeat;
                    v = xyz_cn[((size_t)bi * 3 + a) * n + id] - new_xyz[((size_t)bi * m + p / k) * 3 + a];
                }
                x[((size_t)bi * cin + ci) * mk + p] = v;
            }
}

/* max over the K neighbours (pointnet_utils.py:246), written into a channel slice of y */
EXPORT void oracle_max_over_k(int b, int c, int m, int k, const float *x, float *y, int y_ctotal, int co_off) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int mi = 0; mi < m; ++mi) {
                const float *r = x + (((size_t)bi * c + ci) * m + mi) * k;
                float best = r[0];
                for (int j = 1; j < k; ++j) best = r[j] > best ? r[j] : best;
                y[((size_t)bi * y_ctotal + co_off + ci) * m + mi] = best;
            }
}

/* feature propagation input (pointnet_utils.py:280-294, CUDA semantics: dist = sqrt(d2),
 * pointnet2_utils.py:134): weights 1/(d+1e-8) normalised, then interpolate, then cat([skip, interp]) */
EXPORT void oracle_fp_interpolate_concat(int b, int n, int s, int c1, int c2, const float *unknown,
                                         const float *known, const float *skip, const float *feat_known,
                                         float *out) {
    float *d2 = (float *)malloc((size_t)b * n * 3 * sizeof(float));
    int *id = (int *)malloc((size_t)b * n * 3 * sizeof(int));
    oracle_three_nn(b, n, s, unknown, known, d2, id);
    const int ct = c1 + c2;
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int ci = 0; ci < c1; ++ci)
            memcpy(out + ((size_t)bi * ct + ci) * n, skip + ((size_t)bi * c1 + ci) * n, (size_t)n * sizeof(float));
        for (int p = 0; p < n; ++p) {
            const float *dd = d2 + ((size_t)bi * n + p) * 3;
            const int *ii = id + ((size_t)bi * n + p) * 3;
            float r0 = 1.0f / (sqrtf(dd[0]) + 1e-8f), r1 = 1.0f / (sqrtf(dd[1]) + 1e-8f),
                  r2 = 1.0f / (sqrtf(dd[2]) + 1e-8f);
            float norm = (r0 + r1) + r2;
            float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
            for (int ci = 0; ci < c2; ++ci) {
                const float *row = feat_known + ((size_t)bi * c2 + ci) * s;
                out[((size_t)bi * ct + c1 + ci) * n + p] = (w0 * row[ii[0]] + w1 * row[ii[1]]) + w2 * row[ii[2]];
            }
        }
    }
    free(d2);
    free(id);
}

/* ---- pose fit -------------------------------------------------------------------------------
 * part_fit_st_no_ransac (pose_utils/pose_fit.py:38-53) -> transform_pts_mask
 * (pose_utils/procrustes.py:132-164) with a given rotation, per-point form, double accumulators.
 * sym: transform_pts_2d_mask / rotate_pts_2d_batch (procrustes.py:167-228); the 2x2
 * U diag(1,det(UV^T)) V^T of the cross-covariance M is the rotation by atan2(M10-M01, M00+M11)
 * (identity when that is 0/0, which is what LAPACK's U=V=I gives for M=0). */
static void rot2d_from_cov(const double M[4], double R2[4]) {
    double a = M[0] + M[3], c = M[2] - M[1];
    double h = sqrt(a * a + c * c);
    if (!(h > 0.0)) {
        if (h != h) { R2[0] = R2[1] = R2[2] = R2[3] = NAN; return; }
        R2[0] = 1; R2[1] = 0; R2[2] = 0; R2[3] = 1;
        return;
    }
    double cs = a / h, sn = c / h;
    R2[0] = cs; R2[1] = -sn; R2[2] = sn; R2[3] = cs;
}

EXPORT void oracle_part_fit_st(int b, int p, int n, int sym, const int *labels, const float *src,
                               const float *tgt, int tgt_per_part, const float *rot, const float *given_scale,
                               float *scale, float *trans, int *valid) {
    for (int bi = 0; bi < b; ++bi)
        for (int pi = 0; pi < p; ++pi) {
            const int q = bi * p + pi;
            const float *S = src + (size_t)q * 3 * n;   /* (3,N) predicted NOCS of part pi */
            const float *T = tgt + (size_t)(tgt_per_part ? q : bi) * 3 * n; /* (3,N) camera points */
            const int *lab = labels + (size_t)bi * n;
            double R[9];
            for (int i = 0; i < 9; ++i) R[i] = rot[(size_t)q * 9 + i];
            /* mask = eye[labels] (rows >= P are zero), pose_fit.py:44-45 */
            double cnt = 0, sc[3] = {0, 0, 0}, tc[3] = {0, 0, 0};
            for (int i = 0; i < n; ++i)
                if (lab[i] == pi) {
                    cnt += 1;
                    for (int a = 0; a < 3; ++a) {
                        sc[a] += S[(size_t)a * n + i];
                        tc[a] += T[(size_t)a * n + i];
                    }
                }
            const double den = cnt > 1.0 ? cnt : 1.0; /* clamp(sum(mask), min=1) procrustes.py:137 */
            for (int a = 0; a < 3; ++a) {
                sc[a] /= den;
                tc[a] /= den;
            }
            double Rf[9];
            memcpy(Rf, R, sizeof(R));
            if (sym) {
                /* canon_target = target @ R  (procrustes.py:148) -> 2-D fit on columns (x,z) */
                double c2s[2] = {0, 0}, c2t[2] = {0, 0};
                for (int i = 0; i < n; ++i)
                    if (lab[i] == pi) {
                        double t3[3] = {T[i], T[(size_t)n + i], T[(size_t)2 * n + i]};
                        double ct0 = t3[0] * R[0] + t3[1] * R[3] + t3[2] * R[6];
                        double ct2 = t3[0] * R[2] + t3[1] * R[5] + t3[2] * R[8];
                        c2s[0] += S[i]; c2s[1] += S[(size_t)2 * n + i];
                        c2t[0] += ct0;  c2t[1] += ct2;
                    }
                for (int a = 0; a < 2; ++a) { c2s[a] /= den; c2t[a] /= den; }
                double M[4] = {0, 0, 0, 0}; /* M = tgt_c^T src_c, procrustes.py:168 */
                for (int i = 0; i < n; ++i)
                    if (lab[i] == pi) {
                        double t3[3] = {T[i], T[(size_t)n + i], T[(size_t)2 * n + i]};
                        double ct0 = t3[0] * R[0] + t3[1] * R[3] + t3[2] * R[6] - c2t[0];
                        double ct2 = t3[0] * R[2] + t3[1] * R[5] + t3[2] * R[8] - c2t[1];
                        double s0 = S[i] - c2s[0], s2 = S[(size_t)2 * n + i] - c2s[1];
                        M[0] += ct0 * s0; M[1] += ct0 * s2; M[2] += ct2 * s0; M[3] += ct2 * s2;
                    }
                double R2[4];
                rot2d_from_cov(M, R2);
                /* rot_around_yaxis_to_3d (procrustes.py:69-75), rotation = R @ rot_3d (:151) */
                double R3[9] = {R2[0], 0, R2[1], 0, 1, 0, R2[2], 0, R2[3]};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        double acc = 0;
                        for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * R3[k * 3 + j];
                        Rf[i * 3 + j] = acc;
                    }
            }
            /* scale_pts_mask(source_centered @ R^T, target_centered, w) procrustes.py:117-120,156-158 */
            double num = 0, dn = 0;
            for (int i = 0; i < n; ++i)
                if (lab[i] == pi) {
                    double s3[3], t3[3], rs[3];
                    for (int a = 0; a < 3; ++a) {
                        s3[a] = S[(size_t)a * n + i] - sc[a];
                        t3[a] = T[(size_t)a * n + i] - tc[a];
                    }
                    for (int a = 0; a < 3; ++a) rs[a] = Rf[a * 3] * s3[0] + Rf[a * 3 + 1] * s3[1] + Rf[a * 3 + 2] * s3[2];
                    for (int a = 0; a < 3; ++a) {
                        num += rs[a] * t3[a];
                        dn += rs[a] * rs[a];
                    }
                }
            const double sca = given_scale ? (double)given_scale[q] : num / (dn + 1e-6);
            /* translate_pts_mask(scale * R src, target, w) procrustes.py:123-129,159-162 */
            double tr[3] = {0, 0, 0};
            for (int i = 0; i < n; ++i)
                if (lab[i] == pi) {
                    double s3[3] = {S[i], S[(size_t)n + i], S[(size_t)2 * n + i]};
                    for (int a = 0; a < 3; ++a) {
                        double rs = Rf[a * 3] * s3[0] + Rf[a * 3 + 1] * s3[1] + Rf[a * 3 + 2] * s3[2];
                        tr[a] += (T[(size_t)a * n + i] - sca * rs) / den;
                    }
                }
            scale[q] = (float)sca;
            for (int a = 0; a < 3; ++a) trans[(size_t)q * 3 + a] = (float)tr[a];
            /* valid = count > 3 and finite (pose_fit.py:46, 26-35) */
            float tsum = trans[(size_t)q * 3] + trans[(size_t)q * 3 + 1] + trans[(size_t)q * 3 + 2];
            double rsum = 0;
            for (int i = 0; i < 9; ++i) rsum += R[i];
            valid[q] = (cnt > 3.0) && isfinite(scale[q]) && isfinite(tsum) && isfinite(rsum);
        }
}

/* ---- 3x3 orthogonal Procrustes: rotate_pts_batch (pose_utils/procrustes.py:25-56) ------------
 * R = U diag(1,1,det(U V^T)) V^T for M = tgt^T src = U S V^T.  With v1,v2 the two leading
 * eigenvectors of M^T M and u_i = M v_i / |M v_i|, that matrix is
 * [u1 u2 u1xu2] [v1 v2 v1xv2]^T for either sign of det(M). */
static void jacobi_eig3(double A[9], double V[9]) {
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        for (int pq = 0; pq < 3; ++pq) {
            int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double apq = A[p * 3 + q];
            if (fabs(apq) < 1e-300) continue;
            double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {
                double akp = A[k * 3 + p], akq = A[k * 3 + q];
                A[k * 3 + p] = c * akp - s * akq;
                A[k * 3 + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {
                double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                A[p * 3 + k] = c * apk - s * aqk;
                A[q * 3 + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                V[k * 3 + p] = c * vkp - s * vkq;
                V[k * 3 + q] = s * vkp + c * vkq;
            }
        }
    }
}

static void kabsch3(const double M[9], double R[9]) {
    double A[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += M[k * 3 + i] * M[k * 3 + j];
            A[i * 3 + j] = acc;
        }
    jacobi_eig3(A, V);
    int o[3] = {0, 1, 2};
    double ev[3] = {A[0], A[4], A[8]};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (ev[o[j]] > ev[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
    double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
    for (int k = 0; k < 3; ++k) { v1[k] = V[k * 3 + o[0]]; v2[k] = V[k * 3 + o[1]]; }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1]; v3[1] = v1[2] * v2[0] - v1[0] * v2[2]; v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    double n1 = 0, n2 = 0;
    for (int i = 0; i < 3; ++i) {
        u1[i] = M[i * 3] * v1[0] + M[i * 3 + 1] * v1[1] + M[i * 3 + 2] * v1[2];
        u2[i] = M[i * 3] * v2[0] + M[i * 3 + 1] * v2[1] + M[i * 3 + 2] * v2[2];
        n1 += u1[i] * u1[i]; n2 += u2[i] * u2[i];
    }
    n1 = sqrt(n1); n2 = sqrt(n2);
    for (int i = 0; i < 3; ++i) { u1[i] /= n1; }
    /* re-orthogonalise u2 against u1 (exact in exact arithmetic) */
    double dp = 0;
    for (int i = 0; i < 3; ++i) { u2[i] /= n2; dp += u1[i] * u2[i]; }
    double nn = 0;
    for (int i = 0; i < 3; ++i) { u2[i] -= dp * u1[i]; nn += u2[i] * u2[i]; }
    nn = sqrt(nn);
    for (int i = 0; i < 3; ++i) u2[i] /= nn;
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1]; u3[1] = u1[2] * u2[0] - u1[0] * u2[2]; u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j];
}

EXPORT void oracle_procrustes_rot3(int nb, int n, const float *src, const float *tgt, float *rot) {
    for (int bi = 0; bi < nb; ++bi) {
        const float *S = src + (size_t)bi * n * 3, *T = tgt + (size_t)bi * n * 3;
        double M[9] = {0};
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) M[a * 3 + c] += (double)T[i * 3 + a] * (double)S[i * 3 + c];
        double R[9];
        kabsch3(M, R);
        for (int i = 0; i < 9; ++i) rot[(size_t)bi * 9 + i] = (float)R[i];
    }
}
