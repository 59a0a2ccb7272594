/*
 * captra_hip.h — C ABI of libcaptra_hip.so, the MI355X (gfx950 / CDNA4) implementation of
 * CAPTRA's per-frame point-cloud hot path.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream; nothing here knows
 * about torch.  Each function states which interface of the reference it replaces
 * (paths relative to the reference checkout, network/models/pointnet_lib/src/ unless noted).
 *
 * Conventions
 *   - all tensors are dense, row-major, fp32 / int32, resident in device memory;
 *   - the caller owns every buffer (outputs and scratch included), the callee only writes;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are
 *     asynchronous, no entry point synchronises;
 *   - the return value is a hipError_t as int: 0 on success.  Nothing ever calls exit()
 *     (the reference launchers do, e.g. ball_query_gpu.cu:62-66);
 *   - 64-bit offset arithmetic throughout (the reference uses int and overflows past 2^31
 *     elements).
 */
#ifndef CAPTRA_HIP_H
#define CAPTRA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *captra_stream_t; /* hipStream_t */

/* PER-CALL launch options.  The library holds NO product-affecting state (the reference boundary has none: `extern THCState *state`
 * is declared and unused, pointnet_lib/src/ball_query.cpp:8; every wrapper of pointnet2_api.cpp:10-25 is a pure function of its
 * arguments): what a caller wants different from the defaults travels with the call, as the last argument before the stream of the
 * `_ex` form of an entry point.  NULL, or a zeroed struct, = the defaults = the plain entry point.  Fields an entry point does not
 * read are ignored by it. */
typedef struct captra_launch_opts {
    int splitk_positions; /* dense layers (captra_pointwise_mlp / _mlp2 / _pm / _gn): a launch of at most this many positions (b * l) and
                           * >= 128 input channels splits k over the four waves of a workgroup (partial tiles added in wave order: a
                           * fixed order, 1e-5 relative from the k-ascending chain); 0 = never */
    int sa_prezeroed;     /* SA scales, slice-per-wave form (few clouds): 1 = the caller zeroed the whole output tensor (one fill per
                           * level), 0 = the launcher zeroes its channel slice */
    int centre_m0, centre_mc; /* centre WINDOW: captra_ball_query[_multi], captra_sa_scale_fused (LDS-weights kernels) and
                           * captra_sa_scale_bf16 (small-input scales) process centres [m0, m0 + mc) of every cloud and leave the rest of
                           * their (B, M, ...) outputs untouched (streamed sampling: captra_fps_gather_part); mc <= 0 = all centres.  Entry
                           * points / shapes that cannot honour a window return -2 when one is given */
    int reserved_cus;     /* persistent launches size their grid for this many CUs fewer (another stream's one-workgroup-per-cloud
                           * samplers hold them: every persistent workgroup stays resident) */
    int *dyn_slot;        /* dynamic centre hand-out of the persistent SA kernels: ONE caller-owned device int per launch, zeroed by the
                           * launch on its stream; workgroups that become resident late find the work done.  Same bits.  NULL = static walk */
} captra_launch_opts;

/* ------------------------------------------------------------------------------------------
 * Section 1 — the ten pointnet2_cuda operators (pointnet2_api.cpp:10-25)
 * ---------------------------------------------------------------------------------------- */

/* Replaces furthest_point_sampling_wrapper (sampling.cpp:38-49, kernel sampling_gpu.cu:93-209).
 * xyz (B,N,3) f32; temp (B,N) f32 running min-distance, pre-filled by the caller (1e10,
 * pointnet2_utils.py:27) and left holding the final min distances; idx (B,M) i32.
 * idx[:,0] = 0; round j picks argmax_k temp[k] with strict '>' and LOWEST index on ties
 * (SURVEY.md §8 a1 contract); distance = ((dx*dx + dy*dy) + dz*dz), unfused fp32. */
int captra_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                   captra_stream_t stream);

/* Replaces ball_query_wrapper (ball_query.cpp:14-25, kernel ball_query_gpu.cu:9-45).
 * new_xyz (B,M,3), xyz (B,N,3) f32 -> idx (B,M,nsample) i32: the first `nsample` points k
 * (ascending k) with ((cx-x)^2 + (cy-y)^2) + (cz-z)^2 < radius*radius (strict), remaining
 * slots filled with the first hit; a ball with no hit is written as all zeros (the reference
 * leaves the caller's pre-zeroed buffer untouched, pointnet2_utils.py:261). */
int captra_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                      const float *xyz, int *idx, captra_stream_t stream);
int captra_ball_query_ex(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                      const float *xyz, int *idx, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* Replaces group_points_wrapper (group_points.cpp:25-36, kernel group_points_gpu.cu:47-66).
 * points (B,C,N) f32, idx (B,npoints,nsample) i32 -> out (B,C,npoints,nsample). */
int captra_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                        const int *idx, float *out, captra_stream_t stream);
/* Several grouping jobs over clouds of the same size in ONE launch (all radii x all feature tensors of a set-abstraction level):
 * host arrays of length njobs (channel counts, centres, samples per centre, DEVICE pointers); job j is exactly
 * captra_group_points(b, c[j], n, npoints[j], nsample[j], points[j], idx[j], out[j]).  Falls back to one launch per job for
 * unaligned / odd-sized jobs or more than 12 of them. */
int captra_group_points_multi(int b, int n, int njobs, const int *c, const int *npoints, const int *nsample,
                              const float *const *points, const int *const *idx, float *const *out, captra_stream_t stream);

/* Replaces group_points_grad_wrapper (group_points.cpp:11-22, kernel group_points_gpu.cu:8-25).
 * grad_out (B,C,npoints,nsample), idx -> grad_points (B,C,N) += scatter (caller pre-zeroes): atomicAdd like the reference
 * (this signature has no scratch argument, and the library never allocates). */
int captra_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                             const int *idx, float *grad_points, captra_stream_t stream);
/* The same operator with caller-owned scratch: captra_group_points_grad_ws_bytes() bytes (0 = shape outside the path: C < 8 or
 * N > 16384; pass workspace NULL then).  The index list, shared by every channel, is inverted once into a CSR structure in the
 * scratch and every source point sums its own list in ascending position order: no float atomics, bit-reproducible gradients
 * (csrc/scatter_reduce.hip).  What captra_amd.pointnet2_cuda.group_points_grad_wrapper calls. */
size_t captra_group_points_grad_ws_bytes(int b, int c, int n, int npoints, int nsample);
int captra_group_points_grad_ws(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                                float *grad_points, void *workspace, size_t workspace_bytes, captra_stream_t stream);

/* Replaces gather_points_wrapper (sampling.cpp:11-21, kernel sampling_gpu.cu:8-24).
 * points (B,C,N), idx (B,npoints) -> out (B,C,npoints). */
int captra_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                         float *out, captra_stream_t stream);

/* Replaces gather_points_grad_wrapper (sampling.cpp:24-35, kernel sampling_gpu.cu:46-63). */
int captra_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                              const int *idx, float *grad_points, captra_stream_t stream);
/* The same with caller-owned scratch (atomic-free, bit-reproducible; see captra_group_points_grad_ws).  What
 * captra_amd.pointnet2_cuda.gather_points_grad_wrapper calls. */
size_t captra_gather_points_grad_ws_bytes(int b, int c, int n, int npoints);
int captra_gather_points_grad_ws(int b, int c, int n, int npoints, const float *grad_out, const int *idx, float *grad_points,
                                 void *workspace, size_t workspace_bytes, captra_stream_t stream);

/* Replaces knn_wrapper (interpolate.cpp:26-36, kernel interpolate_gpu.cu:9-57).
 * unknown (B,N,3), known (B,M,3) -> dist2 (B,N,k) squared distances ascending, idx (B,N,k);
 * strict '<' insertion, so the lower index wins ties.  1 <= k <= 200 (reference limit). */
int captra_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
               int *idx, captra_stream_t stream);

/* Replaces three_nn_wrapper (interpolate.cpp:14-24, kernel interpolate_gpu.cu:81-124).
 * unknown (B,N,3), known (B,M,3) -> dist2 (B,N,3) SQUARED distances, idx (B,N,3). */
int captra_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                    int *idx, captra_stream_t stream);

/* Replaces three_interpolate_wrapper (interpolate.cpp:39-53, kernel interpolate_gpu.cu:149-169).
 * points (B,C,M), idx (B,N,3), weight (B,N,3) -> out (B,C,N) = (w0*p0 + w1*p1) + w2*p2. */
int captra_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                             const float *weight, float *out, captra_stream_t stream);

/* Replaces three_interpolate_grad_wrapper (interpolate.cpp:55-68, kernel interpolate_gpu.cu:192-214).
 * grad_out (B,C,N), idx, weight (B,N,3) -> grad_points (B,C,M) += scatter (caller pre-zeroes). */
int captra_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                  const float *weight, float *grad_points, captra_stream_t stream);
/* The same with caller-owned scratch (atomic-free per-known-point sums; see captra_group_points_grad_ws). */
size_t captra_three_interpolate_grad_ws_bytes(int b, int c, int n, int m);
int captra_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                                     float *grad_points, void *workspace, size_t workspace_bytes, captra_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Section 2 — fused operators of the tracking path (no single reference kernel; each replaces
 * a run of ATen ops in network/models/pointnet_utils.py, networks.py, pose_utils/procrustes.py)
 * ---------------------------------------------------------------------------------------- */

/* Canonicalise clouds with the previous pose: out = R^T ((pts + mean) - t) / s
 * (networks.py:38-41 and 184-187).  pts (B,3,N) channel-major as the track loop holds it;
 * mean (B,3); rot (B*P,3,3); trans (B*P,3); scale (B*P); cloud q = b*P + p reads pts[b].
 * Writes out_cn (B*P,3,N) channel-major (network input) and out_n3 (B*P,N,3) point-major
 * (what FPS / ball query / three_nn consume).  Either output may be NULL. */
int captra_canonicalize(int b, int p, int n, const float *pts, const float *mean, const float *rot,
                        const float *trans, const float *scale, float *out_cn, float *out_n3,
                        captra_stream_t stream);
/* The same with a third layout: out_planes (B*P,3,pad256(N)), the clouds in the ball query's LDS plane order (captra_bq_planes),
 * which the level-1 stream kernel's tickets copy into LDS as they are.  Any output may be NULL. */
int captra_canonicalize_planes(int b, int p, int n, const float *pts, const float *mean, const float *rot,
                               const float *trans, const float *scale, float *out_cn, float *out_n3,
                               float *out_planes, captra_stream_t stream);

/* Ball query for up to 4 radii in ONE scan of xyz (PointNetSetAbstractionMsg's loop over
 * radius_list, pointnet_utils.py:228-233).  idx_r (r = 0..nr-1) is (B,M,nsample[r]) i32 with
 * exactly the contents captra_ball_query(radius[r], nsample[r]) would produce.
 * radius, nsample and the idx pointer table are HOST arrays (read at launch time). */
int captra_ball_query_multi(int b, int n, int m, int nr, const float *radius, const int *nsample,
                            const float *new_xyz, const float *xyz, int *const *idx,
                            captra_stream_t stream);
int captra_ball_query_multi_ex(int b, int n, int m, int nr, const float *radius, const int *nsample,
                            const float *new_xyz, const float *xyz, int *const *idx,
                            const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* PACKED WEIGHTS.  The shared-MLP kernels take a layer's weights W^T (cin rows, cout columns; BatchNorm folded in) as ONE
 * buffer of captra_packed_weight_floats(cin, cout) floats holding two images of the same numbers:
 *   [0, KP*CP)   row-major, zero-padded to KP = ceil32(cin) rows x CP = ceil128(cout) columns: every tile a kernel stages is
 *                in bounds and needs no predicate;
 *   [KP*CP, ..)  MFMA-FRAGMENT order for the kernels that stream weights straight into matrix-operand registers: with
 *                KQ = ceil(ceil(cin/2)/4), NT = ceil(cout/32), element ((t*KQ + q)*64 + lane)*4 + i =
 *                W^T[2(4q+i) + (lane>>5)][32t + (lane&31)] (zero beyond cin / cout): one 16-byte load per lane = the A
 *                operands (v_mfma_f32_32x32x2_f32) of four consecutive k-steps of output tile t.  (One dword load per
 *                MFMA kept a CU's vector-memory address unit as busy as its matrix pipes: DESIGN.md section 3.2.)
 * and the bias zero-padded to ceil128(cout).  captra_pack_weights builds both images on device from dense wt (cin,cout) /
 * bias (cout); captra_pack_weights_frag (re)builds the fragment image of a buffer whose row-major image is in place. */
long long captra_packed_weight_floats(int cin, int cout);
int captra_pack_weights(int cin, int cout, const float *wt, const float *bias, float *wt_packed,
                        float *bias_packed, captra_stream_t stream);
int captra_pack_weights_frag(int cin, int cout, float *wt_packed, captra_stream_t stream);

/* One shared-MLP layer, y = act(W x + bias), as an exact-fp32 MFMA GEMM over positions
 * (Conv2d/Conv1d 1x1 + folded BatchNorm + ReLU, pointnet_utils.py:242-245, 296-298).
 *   x (B, cin, L) f32, wt / bias PACKED (see above) -> y (B, cout, L).
 *   act: 0 none, 1 ReLU, 2 sigmoid(x) - 0.5 (networks.py:46).
 * Arithmetic contract: acc = bias; for k in 0..cin-1: acc = fmaf(W[co][k], x[k][l], acc)
 * (what v_mfma_f32_32x32x2_f32 computes), then the activation. */
int captra_pointwise_mlp(int b, int cin, int cout, long long l, const float *x, const float *wt,
                         const float *bias, int act, float *y, captra_stream_t stream);
int captra_pointwise_mlp_ex(int b, int cin, int cout, long long l, const float *x, const float *wt,
                         const float *bias, int act, float *y, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */
/* The same layer with a bias PER CLOUD, bias_bc (B, ceil128(cout)) f32, zero beyond cout: y[b] = act(W x[b] + bias_bc[b]).
 * For a layer on [x; repeat(v)] with one vector v per cloud (pointnet_utils.py:265-270) after the caller has formed
 * bias_bc[b] = W2 v[b] + bias (captra_pointwise_mlp with l = 1): a third of the products, but not the k-ascending chain over the
 * concat -- the f32x6 mode uses it, the exact mode keeps captra_pointwise_mlp2.  -2: shape outside the direct kernels. */
int captra_pointwise_mlp_cb(int b, int cin, int cout, long long l, const float *x, const float *wt,
                            const float *bias_bc, int act, float *y, captra_stream_t stream);

/* First layer of a set-abstraction scale with the group-and-concat fused into the operand load
 * (group_operation + "-= centre" + cat, pointnet_utils.py:234-240):
 *   x[b][ci][m][k] = feat[b][ci][idx[b][m][k]]                 for ci <  cfeat
 *                  = xyz_cn[b][ci-cfeat][idx] - new_xyz[b][m]   for ci >= cfeat   (3 channels)
 * feat (B,cfeat,N) or NULL when cfeat == 0; xyz_cn (B,3,N); new_xyz (B,M,3); idx (B,M,K).
 * y (B,cout,M,K) = relu(W x + bias), same arithmetic contract as captra_pointwise_mlp. */
int captra_sa_group_mlp(int b, int n, int m, int k, int cfeat, int cout, const float *feat,
                        const float *xyz_cn, const float *new_xyz, const int *idx, const float *wt,
                        const float *bias, float *y, captra_stream_t stream);

/* Last layer of a set-abstraction scale with the max over the K neighbours fused into the
 * epilogue (pointnet_utils.py:245-246): x (B,cin,M,K) -> y[b][co_off+co][m] =
 * max_k relu(W x + bias), y being (B, y_ctotal, M) so that the scales of one SA level write
 * straight into the concatenated tensor (pointnet_utils.py:249). */
int captra_mlp_max(int b, int cin, int cout, int m, int k, const float *x, const float *wt,
                   const float *bias, float *y, int y_ctotal, int co_off, captra_stream_t stream);

/* One whole set-abstraction scale in one launch: gather + centre-subtract + concat, three shared-MLP
 * layers and the max over the K neighbours (the loop body of PointNetSetAbstractionMsg.forward,
 * pointnet_utils.py:228-248).  Neither the grouped tensor nor the intermediate activations touch HBM.
 * Shapes as captra_sa_group_mlp / captra_mlp_max; w_i / b_i PACKED (see captra_pack_weights);
 * k in {32, 64, 128}, c_i <= 256.  out (B,out_ctotal,M) receives channels [co_off, co_off + c3).
 * Bit-identical to captra_sa_group_mlp -> captra_pointwise_mlp -> captra_mlp_max. */
int captra_sa_scale_fused(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *feat,
                          const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                          const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                          float *out, int out_ctotal, int co_off, captra_stream_t stream);
int captra_sa_scale_fused_ex(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *feat,
                          const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                          const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                          float *out, int out_ctotal, int co_off, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* Dense layer inside a Conv1d -> GroupNorm -> ReLU chain (the rotation heads, blocks.py:150-165) without the separate
 * normalisation pass.  As captra_pointwise_mlp, plus:
 *   ab_in     (B,cin,2) or NULL: the input tensor is the previous layer's RAW output; every element is read as
 *             relu(a*x + b) with that layer's per-(cloud, channel) GroupNorm coefficients;
 *   stats_out (B,cout,stats_t,2) or NULL: per output channel and 64-position tile, (sum, sum of squares) of this layer's
 *             raw output (act must be CAPTRA_ACT_NONE); needs cout > 64 (-2 otherwise); stats_t = 2*ceil(l/128), or
 *             2*ceil(l/64) (32-position tiles) when ceil(l/64)*ceil(cout/64)*b < 2048 -- captra_pointwise_mlp_gn_tiles().
 * captra_gn_finalize reduces the partials (fixed order, double precision) to ab (B,c,2): a = gamma*rstd, b = beta - mean*a,
 * mean / biased variance over the group's channels_per_group channels x n positions (torch.nn.GroupNorm semantics).
 * Same convolution bits as captra_pointwise_mlp; the normalisation differs from a two-pass GroupNorm by rounding only. */
int captra_pointwise_mlp_gn(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                            const float *bias_packed, const float *ab_in, int act, float *y, float *stats_out,
                            int stats_t, captra_stream_t stream);
int captra_pointwise_mlp_gn_ex(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                            const float *bias_packed, const float *ab_in, int act, float *y, float *stats_out,
                            int stats_t, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */
int captra_pointwise_mlp_gn_tiles(int b, int cout, long long l);
int captra_pointwise_mlp_gn_tiles_ex(int b, int cout, long long l, const captra_launch_opts *opts);   /* (the split-k form has its own tile size) */   /* the stats_t captra_pointwise_mlp_gn expects for this shape */
int captra_gn_finalize(int b, int c, int channels_per_group, int stats_t, long long n, float eps, const float *stats,
                       const float *gamma, const float *beta, float *ab, captra_stream_t stream);

/* bf16-operand / fp32-accumulate variants of the shared-MLP kernels (BASELINE.json configs[2]; opt-in, the default path is
 * exact fp32): y = act(b + sum_k bf16(w[k]) * bf16(x[k])), weights rounded (RNE) once by captra_pack_weights_bf16 into
 * wb [ceil32(cout)][ceil32(cin)] bf16 (UNtransposed, k contiguous, zero padded) from dense wt (cin,cout) fp32; every layer's
 * input is rounded when it becomes an MFMA operand; bias (the fp32 packed bias of captra_pack_weights) and accumulation
 * in fp32; tensors in memory stay fp32.
 *   captra_pointwise_mlp_bf16: as captra_pointwise_mlp.
 *   captra_pointwise_mlp_bf16_pm: the same layer with y POINT-major (B,L,cout) fp32, cout % 4 == 0.
 *   captra_sa_scale_bf16 (csrc/sa_bf16.hip): one SA scale, register-resident, from a weight IMAGE built once per scale by
 *     captra_pack_sa_bf16 (captra_sa_bf16_image_bytes bytes) from the three layers' packed fp32 buffers: MFMA-fragment-ordered
 *     bf16 weights (layers 2 / 3 with the k order the in-register hand-over produces) followed by the fp32 biases of layers
 *     2 / 3.  pre = 0: feat_or_v1 = feat (B,cfeat,N) fp32, cfeat + 3 <= 6, b1 rides as two constant-one input rows (hi + lo
 *     bf16 split); pre = 1: feat_or_v1 = v1 (B,N,c1) fp32 POINT-major = b1 + W1[feature rows] feat (captra_pointwise_mlp_bf16_pm
 *     on the layer's leading rows), the image's first layer holds the three xyz rows only.  Layer 3's bias is added after the
 *     max over the neighbours (equal to adding it before: rounding is monotone).  Instantiated for the CAPTRA backbone shapes
 *     (same list as captra_sa_scale_fused's register-resident kernels, k = 32 / 64 / 128 as in the configs); -2 otherwise. */
int captra_pack_weights_bf16(int cin, int cout, const float *wt, unsigned short *wb, captra_stream_t stream);
int captra_pointwise_mlp_bf16(int b, int cin, int cout, long long l, const float *x, const unsigned short *wb,
                              const float *bias_packed, int act, float *y, captra_stream_t stream);
int captra_pointwise_mlp_bf16_pm(int b, int cin, int cout, long long l, const float *x, const unsigned short *wb,
                                 const float *bias_packed, int act, float *y, captra_stream_t stream);
long long captra_sa_bf16_image_bytes(int cfeat, int c1, int c2, int c3);
int captra_pack_sa_bf16(int cfeat, int c1, int c2, int c3, int pre, const float *wt1_packed, const float *b1_packed,
                        const float *wt2_packed, const float *b2_packed, const float *wt3_packed, const float *b3_packed,
                        unsigned char *img, captra_stream_t stream);
int captra_sa_scale_bf16(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                         const float *xyz_cn, const float *new_xyz, const int *idx, const unsigned char *img, float *out,
                         int out_ctotal, int co_off, captra_stream_t stream);
int captra_sa_scale_bf16_ex(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                         const float *xyz_cn, const float *new_xyz, const int *idx, const unsigned char *img, float *out,
                         int out_ctotal, int co_off, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* f32x6: fp32-EQUIVALENT shared MLPs on the bf16 matrix pipe (opt-in arithmetic `mlp_dtype = "f32x6"`; the default path stays the
 * exact fp32 fmaf chain).  Replaces, like the kernels above, the fp32 Conv2d 1x1 + BN + ReLU stacks of
 * network/models/pointnet_utils.py:242-249 and blocks.py:118-135,168-193.  Every operand v is held as three bf16 numbers
 * v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1) (RNE; v = v0 + v1 + v2 exactly for normal fp32 v) and a layer is
 *     y = act(b + sum_k [w0 x0 + w0 x1 + w1 x0 + w1 x1 + w0 x2 + w2 x0]_k),
 * six v_mfma_f32_32x32x16_bf16 per 16-wide k-step, products exact, fp32 accumulation: the three dropped products are <= 2^-24 |w x|
 * each, so results differ from the exact chain by fp32-roundoff-sized terms (<= 2e-6 of a layer's largest output,
 * tests/test_x6_gpu.py) -- NOT bit for bit -- at 16 / 6 of the fp32 matrix-pipe rate.
 *   captra_sa_scale_x6 (csrc/sa_x6.hip): one SA scale, register-resident, as captra_sa_scale_bf16.  img: built once per scale by
 *     captra_pack_sa_x6 (captra_sa_x6_image_bytes bytes) from the packed fp32 buffers of layers 2 / 3 (split fragment triples, k order
 *     of the in-register hand-over) followed by the fp32 biases b1 (zeros when b1_packed is NULL), b2, b3.  w1_packed: the first
 *     layer's packed fp32 buffer -- its 3..6 xyz / feature rows run on the exact v_mfma_f32_32x32x2_f32.  pre = 0: feat_or_v1 = feat
 *     (B,cfeat,N) fp32 or NULL (cfeat = 0), cfeat + 3 <= 6; pre = 1: feat_or_v1 = v1 (B,N,c1) fp32 POINT-major = b1 + W1[feature rows] feat
 *     (captra_pointwise_mlp_pm, exact).  k % 32 == 0.  Instantiated for the CAPTRA backbone shapes; -2 otherwise. */
long long captra_sa_x6_image_bytes(int cfeat, int c1, int c2, int c3);
int captra_pack_sa_x6(int cfeat, int c1, int c2, int c3, const float *b1_packed, const float *wt2_packed, const float *b2_packed,
                      const float *wt3_packed, const float *b3_packed, unsigned char *img, captra_stream_t stream);
int captra_sa_scale_x6(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, int pre, const float *feat_or_v1,
                       const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1_packed, const unsigned char *img,
                       float *out, int out_ctotal, int co_off, captra_stream_t stream);

/*   captra_pointwise_mlp_x6 (csrc/dense_x6.hip): captra_pointwise_mlp_gn in the f32x6 arithmetic -- x (B,cin,L) fp32 read as
 *     relu(fmaf(a, x, b)) when ab_in (B,cin,2) is given, y (B,cout,L) = act(W x' + bias), stats_out (B,cout,stats_t,2) or NULL: (sum, sum of
 *     squares) of the raw output per channel and 128 positions, stats_t = captra_pointwise_mlp_x6_tiles(l) = l / 128 (act must be
 *     CAPTRA_ACT_NONE then) -- the layout captra_gn_finalize reduces.  wimg: captra_pack_dense_x6 of the layer's packed fp32 buffer
 *     (captra_dense_x6_image_bytes bytes).  Shapes: cin % 16 == 0, cin <= 1024, cout % 256 == 0, l % 128 == 0 (the rotation heads' 128 ->
 *     512 -> 512 -> 256 on 4096-point clouds, blocks.py:168-193); -2 otherwise (the caller runs captra_pointwise_mlp_gn). */
long long captra_dense_x6_image_bytes(int cin, int cout);
int captra_pack_dense_x6(int cin, int cout, const float *wt_packed, unsigned char *img, captra_stream_t stream);
int captra_pointwise_mlp_x6_tiles(long long l);
int captra_pointwise_mlp_x6(int b, int cin, int cout, long long l, const float *x, const unsigned char *wimg, const float *bias_packed,
                            const float *ab_in, int act, float *y, float *stats_out, int stats_t, captra_stream_t stream);

/* f32x6 dense CHAINS (csrc/chain_x6.hip): captra_mlp_chain3 / captra_coord_tail in the f32x6 arithmetic, c0 in {131, 134} -> 128 -> 128
 * -> 128 [-> seg_dim; -> 128 -> nocs_dim].  img: captra_chain_x6_image_bytes(nl, cin[], cout[]) bytes, filled layer by layer with
 * captra_pack_chain_x6(l, cin, cout, natural = (l == 0), frag_off = bytes of the fragments of the layers before l, total_frag_bytes,
 * the layer's packed fp32 weights and bias) -- split fragment triples of every layer, then 128 fp32 bias slots per layer. */
long long captra_chain_x6_image_bytes(int nl, const int *cin, const int *cout);
int captra_pack_chain_x6(int l, int cin, int cout, int natural, long long frag_off, long long total_frag_bytes, const float *wt_packed,
                         const float *bias_packed, unsigned char *img, captra_stream_t stream);
int captra_mlp_chain3_x6(int b, int c0, long long l, const float *x, const unsigned char *img, int act3, float *y, captra_stream_t stream);
int captra_coord_tail_x6(int b, int c0, int seg_dim, int nocs_dim, long long l, const float *x, const unsigned char *img, int nocs_act,
                         float *seg, float *nocs, captra_stream_t stream);

/* bf16-NATIVE dense layers (csrc/dense_bf16.hip): activations in HBM as bf16, POINT-major (B,L,ceil32(C)), channels in SLOT
 * ORDER (inside every aligned block of 16 channels memory slot s holds channel perm[s] = {0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15},
 * the order in which a 32x32 MFMA accumulator tile hands its rows to a lane; padding channels are zero).  Same per-layer
 * contract as captra_pointwise_mlp_bf16.  The weights come as a fragment image built once by captra_pack_dense_bf16 from the
 * layer's packed fp32 buffer (perm = 1 when the layer's INPUT is such a tensor, 0 for a channel-major fp32 input).
 *   captra_pointwise_mlp_bf16pm: in_pm / out_pm choose the layouts of x / y ((B,cin,L) / (B,cout,L) fp32 when 0); ab != NULL
 *     (in_pm only): (B,cin,2) GroupNorm coefficients of the layer that produced x, applied as relu(a x + b) while the operand is
 *     loaded (blocks.py:150-165's GroupNorm + ReLU without a pass of its own).
 *   captra_gn_stats_bf16pm: per-channel partial (sum, sum of squares) of a STORED tensor over chunks of 128 positions ->
 *     stats (B,c,T,2), T = captra_gn_stats_bf16pm_tiles(l); captra_gn_finalize turns them into ab.
 *   captra_pointwise_mlp_bf16pm_stats: the layer with a point-major output that ALSO leaves those partial statistics of what it
 *     stored, from its own epilogue (no second pass over the tensor): stats (B,T,cout,2) TILE-major, T =
 *     captra_dense_bf16_stats_tiles(l) (chunks of 64 positions; fixed summation order, no atomics); captra_gn_finalize_tm
 *     is captra_gn_finalize for that layout. */
long long captra_dense_bf16_image_bytes(int cin, int cout);
int captra_pack_dense_bf16(int cin, int cout, int perm, const float *wt_packed, unsigned char *img, captra_stream_t stream);
int captra_pointwise_mlp_bf16pm(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg,
                                const float *bias_packed, const float *ab, int act, int out_pm, void *y, captra_stream_t stream);
/* ... with a bias per cloud, bias_per_cloud (B,cout) fp32, cout % 32 == 0: a feature-propagation layer whose interpolated input
 * is ONE vector per cloud (pointnet_utils.py:265-268, S == 1: the repeat + concat + 1x1 conv is W1 x + (W2 v + b)). */
int captra_pointwise_mlp_bf16pm_cb(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg,
                                   const float *bias_per_cloud, int act, int out_pm, void *y, captra_stream_t stream);
int captra_gn_stats_bf16pm_tiles(long long l);
int captra_dense_bf16_stats_tiles(long long l);
int captra_gn_finalize_tm(int b, int c, int channels_per_group, int stats_t, long long n, float eps, const float *stats,
                          const float *gamma, const float *beta, float *ab, captra_stream_t stream);
int captra_pointwise_mlp_bf16pm_stats(int b, int cin, int cout, long long l, int in_pm, const void *x, const unsigned char *wimg,
                                      const float *bias_packed, const float *ab, int act, void *y, float *stats,
                                      captra_stream_t stream);
int captra_gn_stats_bf16pm(int b, int c, long long l, const void *x, float *stats, captra_stream_t stream);

/* bf16-native dense layers, LDS-tiled (csrc/tile_bf16.hip; round 4): the same per-layer contract on point-major tensors, with the
 * operand tile (128 positions x 128 channels per K-chunk) staged through LDS -- the producer's GroupNorm applied once per element
 * while it is staged -- and every weight fragment feeding four position tiles.  Replaces, inside the rotation heads
 * (blocks.py:147-193) and the backbone's point-major layers, the streaming kernels above where the shape is instantiated.
 *   captra_dense_bf16_tile: x (B,L,ceil32(cin)) -> y (B,L,ceil32(cout)), both bf16 slot order; wimg packed with perm = 1; bias_bs = 0
 *     (one bias) or cout (bias (B,cout), cout % 32 == 0); ab (B,cin,2) or NULL; act CAPTRA_ACT_NONE / RELU; stats (B,T,cout,2) or
 *     NULL: partial (sum, sum of squares) of the layer's fp32 outputs per chunk of 128 positions, T =
 *     captra_dense_bf16_tile_stats_tiles(l), for captra_gn_finalize_tm.  Returns -2 for cout < 64 (use captra_pointwise_mlp_bf16pm).
 *   captra_head12_bf16: layers 1 + 2 of a Conv -> GroupNorm -> ReLU head, cin <= 128 -> 512 -> 512, without y1 ever reaching HBM.
 *     ab1 == NULL: the statistics pass (stats <- y1's partial sums; w2img, bias2, y2 unused).  ab1 != NULL: y1 is recomputed,
 *     normalised in registers, parked in LDS as the second layer's operand; y2 (B,L,512) raw bf16 and stats <- y2's partial sums. */
int captra_dense_bf16_tile_stats_tiles(long long l);
int captra_dense_bf16_tile(int b, int cin, int cout, long long l, const void *x, const unsigned char *wimg, const float *bias_packed,
                           long long bias_bs, const float *ab, int act, void *y, float *stats, captra_stream_t stream);
/* captra_dense_bf16_tile_ex: the same kernel with a channel-major fp32 input and / or an fp32 or pooled output (the 128- and
 * 512-point levels of the backbone: SA3's group_all MLP + max, FP3, FP2 -- pointnet_utils.py:253-343).  in_cm = 1: x (B,csplit,L)
 * fp32 holds input channels [0, csplit), x2 (B,cin - csplit,L) the rest (csplit = cin, x2 = NULL for one tensor): the
 * [xyz, feat] concat of sample_and_group_all (pointnet_utils.py:171-188) is never built; l % 4 == 0.  out_mode 0: y (B,L,ceil32(cout))
 * bf16 slot order; 1: y (B,cout,L) fp32; 2: y (B,cout) fp32 = max over the l <= 128 positions (torch.max(-1) of pointnet_utils.py:341).
 * wimg packed with perm = 1 in every case. */
int captra_dense_bf16_tile_ex(int b, int cin, int cout, long long l, int in_cm, const void *x, const float *x2, int csplit,
                              const unsigned char *wimg, const float *bias_packed, long long bias_bs, const float *ab, int act,
                              int out_mode, void *y, float *stats, captra_stream_t stream);
/* One vector per cloud through a layer (FP3's per-cloud bias W2 v + b, pointnet_utils.py:265-268 with S == 1): x (B,cin) fp32 ->
 * y (B,cout) fp32 = bias + sum_k bf16(w[k]) bf16(x[k]); wt_packed / bias_packed as for captra_pointwise_mlp. */
int captra_gemv_bf16(int b, int cin, int cout, const float *x, const float *wt_packed, const float *bias_packed, float *y,
                     captra_stream_t stream);
int captra_head12_bf16(int b, int cin, long long l, const void *x, const unsigned char *w1img, const float *bias1_packed,
                       const float *ab1, const unsigned char *w2img, const float *bias2_packed, void *y2, float *stats,
                       captra_stream_t stream);
int captra_head12_bf16_ex(int b, int cin, long long l, const void *x, const unsigned char *w1img, const float *bias1_packed,
                       const float *ab1, const unsigned char *w2img, const float *bias2_packed, void *y2, float *stats,
                       const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* bf16 mode, register-resident dense CHAIN (csrc/sa_bf16.hip): FP1's shared MLP + the backbone's conv1 (+ CoordinateNet's two heads)
 * in one launch.  x (B,c0,L) fp32, c0 <= 144 -> three 128-wide Conv+BN+ReLU layers; feat_pm (B,L,128) bf16 slot order receives the
 * result when non-NULL; heads != 0: seg (B,s,L) = Ws feat + bs and nocs (B,no,L) = sigmoid(Wo relu(Wh feat + bh) + bo) - 0.5
 * (s, no <= 32).  img (captra_chain_bf16_image_bytes bytes): the layers' captra_pack_dense_bf16 images back to back -- trunk
 * layer 1 with perm = 0, every other layer with perm = 1; order trunk 1-3, seg, hidden, out -- followed by each layer's bias
 * as 32 floats per row tile (zero padded), same order. */
long long captra_chain_bf16_image_bytes(int c0, int heads);
int captra_mlp_chain_bf16(int b, int c0, long long l, int heads, int s, int no, const float *x, const unsigned char *img,
                          void *feat_pm, float *seg, float *nocs, captra_stream_t stream);

/* Furthest point sampling + index_points in one launch (pointnet_utils.py:222-223: new_xyz = index_points(xyz,
 * farthest_point_sample(xyz, S))): xyz (B,N,3) -> idx (B,M) i32, new_xyz_n3 (B,M,3), new_xyz_cn (B,3,M) (either output
 * pointer may be NULL).  Same selection rule as captra_furthest_point_sampling with temp = 1e10.  Returns -2 when the
 * cloud does not fit the register-resident kernel (N > ~32k): sample and gather separately then. */
int captra_fps_gather(int b, int n, int m, const float *xyz, int *idx, float *new_xyz_n3, float *new_xyz_cn,
                      captra_stream_t stream);

/* Candidate extraction of the on-the-fly ball crop (reference nocs_data_process.py:92-109, 151-163; nocs_utils.py:5-33), one
 * workgroup per tracked instance: inside the inclusive image box box[i] = {row_min, col_min, row_max, col_max} back-project the
 * pixels with depth > 0 (float64: ray = kinv (col, h - row, 1); p = ray * z / ray_z; (p_x, p_y, -p_z) * 0.001) and keep those with
 * |p - center[i]| <= radius[i], in row-major pixel order: pts (B,cap,3) float64, obj (B,cap) = the instance mask at those pixels,
 * pix (B,cap) = row * w + col, counts (B,2) = {members found (may exceed cap: only the first cap are stored), pixels with
 * depth > 0 in the box}.  depth (B,h,w) int32 millimetres, mask (B,h,w) bytes, kinv 9 doubles (row-major inverse intrinsics). */
int captra_crop_ball(int b, int h, int w, int cap, const int *depth, const unsigned char *mask, const int *box,
                     const double *center, const double *radius, const double *kinv, double *pts, unsigned char *obj,
                     int *pix, int *counts, captra_stream_t stream);

/* The crop's inputs from the DEVICE-resident pose of the previous frame (reference nocs_data_process.py:136-148 `proj_corners`, model.py:425-452):
 * trans (B,3), scale (B) fp32 -> center (B,3) = double(trans), radius (B) = max(radius_factor * double(scale), 0.05), box (B,4) = the inclusive,
 * clamped image box {row_min, col_min, row_max, col_max} of the cube c +- r under the intrinsics kmat (9 doubles, row-major), in float64
 * with numpy's operation order and int32 truncation.  What captra_crop_ball reads: no host round trip between the pose and the crop. */
int captra_crop_box(int b, int h, int w, double radius_factor, const float *trans, const float *scale, const double *kmat, int *box,
                    double *center, double *radius, captra_stream_t stream);
/* The rest of the re-crop without the host (csrc/crop.hip; reference nocs_data_process.py:92-109, 43-50, 227-236): the candidate
 * lists of the crops (member table repeated until >= num_points entries) as the sampler's fp32 input with their lengths, from the
 * DEVICE-resident member counts of captra_crop_ball -- cand (B,stride,3), lens (B,) for captra_fps_gather_ragged, info[4] (zeroed
 * by the call): [0] != 0 when an instance is on a rare path (< 10 members: the crop's radius grows; a list longer than stride <= 5
 * num_points: thinning by the host's generator) whose frame the caller runs again with the host in the loop, [1] = longest list --
 * and the sampler's picks turned into the frame's tensors in the networks' layouts: points - mean (B,3,n) fp32, labels (B,n)
 * int64, ground-truth NOCS (B,3,n) fp32 of the object's points (float64 arithmetic, rot (B,3,3), trans (B,3), scale (B,) float64). */
int captra_otf_candidates(int b, int cap, int stride, int num_points, const double *pts, const int *counts, float *cand, int *lens,
                          int *info, captra_stream_t stream);
int captra_otf_finish(int b, int cap, int stride, int n, const double *pts, const unsigned char *obj, const int *counts,
                      const int *picks, const float *mean, const double *rot, const double *trans, const double *scale,
                      float *points_cn, long long *labels, float *nocs_cn, captra_stream_t stream);

/* Ragged batch of the same operation: the clouds are padded to n_stride points each (xyz (B,n_stride,3)) and cloud i
 * samples m of its FIRST n_per_cloud[i] points (device array of B ints, 1 <= n_per_cloud[i] <= n_stride; NULL = all
 * n_stride).  This is the re-sampling step of the on-the-fly ball crop (reference datasets/data_utils.py:138-157:
 * farthest_point_sample of up to 5 x num_points candidates, a different count per tracked instance), batched over the
 * trajectories of a step.  Clouds of 8192-20480 points take the spatially pruned kernel (csrc/fps_pruned.hip), picks
 * identical to captra_furthest_point_sampling. */
int captra_fps_gather_ragged(int b, int n_stride, const int *n_per_cloud, int m, const float *xyz, int *idx,
                             float *new_xyz_n3, float *new_xyz_cn, captra_stream_t stream);

/* STREAMED sampling: picks [j0, j1) of captra_fps_gather's m as a launch of its own.  state (B,N) fp32 carries a cloud's running
 * minima from part to part (every part writes it; every part but the one starting at 0 reads it); idx / new_xyz_* are the whole
 * sampling's buffers.  Parts launched in order on one stream produce the one launch's picks bit for bit (sampling_gpu.cu:93-209
 * is one loop; this cuts it at j0).  Between two parts the caller may run whatever needs only the centres picked so far -- the
 * ball query and the shared MLPs of those centres, with captra_launch_opts::centre_m0 / centre_mc -- on other streams, so that the sampler's
 * dependent rounds, one workgroup per cloud, no longer stand alone at the head of a frame.  -2: cloud outside the
 * register-resident kernel (> 8191 points). */
int captra_fps_gather_part(int b, int n, int m, int j0, int j1, const float *xyz, float *state, int *idx, float *new_xyz_n3,
                           float *new_xyz_cn, captra_stream_t stream);

/* A module of the backbone's NECK in the bf16 mode as ONE launch (csrc/neck_bf16.hip): nl = 2 or 3 layers act(b + W x), ReLU between them,
 * on tiles of 64 positions with every hidden activation in LDS as the next layer's operand image and the weights streamed from the
 * layers' captra_pack_dense_bf16(perm = 1) images wimg[i] (+ packed fp32 biases bias[i]); channel counts c[0] (input) .. c[nl], hidden
 * widths multiples of 128 up to 256 then 512.  Layer 1's input (B,c[0],L) is never built: channels [0, csplit) come from x (B,csplit,L) and
 *   kind 0 -- SA3, PointNetSetAbstraction with group_all (pointnet_utils.py:302-343): the rest from x2 (B,c[0]-csplit,L); y (B,c[nl]) fp32 =
 *            act_last(max over the L positions) (zeroed by the call on `stream`; act_last must be CAPTRA_ACT_RELU);
 *   kind 1 -- FP3, PointNetFeaturePropagation with one source vector per cloud (pointnet_utils.py:265-298): csplit = c[0]; layer 1's bias
 *            of cloud b is bias[0] + W_v bf16(v[b]), v (B,cv) fp32, gw (cv, ceil128(c[1])) bf16 = the RNE rounding of the packed fp32 W'^T rows
 *            that multiply v (captra_gemv_bf16 rounds the same rows on the fly: same arithmetic); y (B,c[nl],L) fp32;
 *   kind 2 -- FP2, PointNetFeaturePropagation (pointnet_utils.py:280-298): the rest interpolated from x2 (B,c[0]-csplit,S) through nn_idx /
 *            nn_w (B,L,3) of captra_three_nn_weights, (w0 f[j0] + w1 f[j1]) + w2 f[j2] as captra_interp_concat; y (B,c[nl],L) fp32.
 * Bit-identical to the chain of captra_dense_bf16_tile_ex launches (+ captra_gemv_bf16 / captra_interp_concat) it replaces.  -2: shapes
 * outside the kernel. */
int captra_neck_chain_bf16(int kind, int b, long long l, int nl, const int *c, const float *x, const float *x2, int csplit,
                           const unsigned char *const *wimg, const float *const *bias, const int *nn_idx, const float *nn_w, int s_known,
                           const float *v, const unsigned short *gw, int cv, int act_last, float *y, captra_stream_t stream);

/* QueryAndGroup(radius, nsample, use_xyz)(xyz, new_xyz, features) of the reference (pointnet_lib/pointnet2_utils.py:274-310: ball_query
 * -> grouping_operation -> centre subtraction -> cat) in ONE launch: out (B, C + 3, M, K) = cat([features[:, :, idx], xyz[idx] -
 * new_xyz]) -- features first; (B,3,M,K) when features == NULL, (B,C,M,K) when use_xyz == 0 -- with idx the ball query's lists
 * (captra_ball_query, bit for bit; also written to idx_out (B,M,K) when non-NULL).  xyz (B,N,3), new_xyz (B,M,3), features (B,C,N).
 * The lists stay in LDS and the cloud is staged once per workgroup for the search AND the coordinate channels: nothing goes through
 * HBM between the two ops.  -2: nsample % 4 != 0, out / idx_out not 16-byte aligned, N > 8192 (captra_ball_query +
 * captra_group_points then). */
int captra_query_and_group(int b, int n, int m, float radius, int nsample, int c, int use_xyz, const float *xyz, const float *new_xyz,
                           const float *features, float *out, int *idx_out, captra_stream_t stream);

/* LEVEL-1 STREAM (csrc/sa_bf16.hip, bf16 mode): everything PointNetSetAbstractionMsg.forward (pointnet_utils.py:214-249) does at the
 * first level of PointNet2Msg, for the (one or two) networks that share a cloud, in ONE launch -- furthest point sampling
 * (sampling_gpu.cu:93-209; fps_idx (B,M), new_xyz in both layouts), the three ball queries (ball_query_gpu.cu:9-45; idx3[s]
 * (B,M,K_s), K = 32 / 64 / 128, radius3[s]) and the pooled features out_a / out_b (B,320,M) of the CAPTRA SA1 shapes
 * [cf+3 -> 32 -> 32 -> 64], [-> 64 -> 64 -> 128], [-> 64 -> 96 -> 128] on feat_a (B,cfa,N) / feat_b (B,cfb,N), cf in {0, 3} (feat NULL for
 * 0; cfb < 0: one network).  B workgroups run the sampler's dependent rounds and publish every 32 picks; all other workgroups
 * consume (window, scale, cloud) tickets -- ball query of the window's centres, then the scale's shared MLPs -- while the sampler
 * picks on, so the 0.24 ms it used to hold the step alone are filled.  Every output equals captra_fps_gather +
 * captra_ball_query_multi + 3 x captra_sa_scale_bf16 per network, bit for bit.  img_*3[s]: the scales' captra_pack_sa_bf16 images
 * (pre = 0).  scratch: captra_sa1_stream_scratch_bytes(b, m) bytes, zeroed by the call on `stream`; once the launch completed,
 * word 1 of the 16 unsigned words at scratch + 8*b*m is non-zero iff a consumer gave up waiting (bounded spins; outputs incomplete
 * then).  m2 > 0 (<= 256): the sampler workgroups go on to pick m2 of their m centres -- the second level's sampling
 * (captra_fps_gather on new_xyz_n3, bit for bit: fps2_idx (B,m2) indexes the level-1 centres, new2_xyz_* their coordinates) -- while
 * the consumers work off their backlog.  planes (optional, NULL = staged from xyz_n3 by every ticket): the clouds once more, (B,3,pad256(N)) in the ball query's LDS
 * plane order (captra_bq_planes: element ((chunk / 4) * 64 + lane) * 4 + chunk % 4 of plane a = coordinate a of point 64 chunk + lane, +inf
 * beyond N), so that a ticket's staging is straight 16-byte copies.  -2: n > 4096, m > 512, m % 32 != 0, b > 256 or feature counts
 * outside {0, 3}. */
long long captra_sa1_stream_scratch_bytes(int b, int m);
int captra_bq_planes(int b, int n, const float *xyz_n3, float *planes, captra_stream_t stream);
int captra_sa1_stream_bf16(int b, int n, int m, const float *xyz_n3, const float *xyz_cn, const float *planes, const float *radius3, int *fps_idx,
                           float *new_xyz_n3, float *new_xyz_cn, int *const *idx3, int cfa, const float *feat_a,
                           const unsigned char *const *img_a3, float *out_a, int cfb, const float *feat_b,
                           const unsigned char *const *img_b3, float *out_b, int m2, int *fps2_idx, float *new2_xyz_n3,
                           float *new2_xyz_cn, void *scratch, captra_stream_t stream);

/* SA scale with a pre-transformed first layer.  Layer 1's k-ascending chain runs over the cfeat feature rows first and the
 * three relative-xyz rows last (pointnet_utils.py:234-240), and its first cfeat steps depend on the SOURCE point only:
 *   v1 (B,c1,N) = captra_pointwise_mlp(feat (B,cfeat,N), w1 rows 0..cfeat-1, b1, CAPTRA_ACT_NONE)      (once per source point)
 * The kernel gathers v1[:, idx] as the accumulator start and continues the chain with the xyz rows of w1 (same packed
 * buffer), then layers 2, 3 and the max as captra_sa_scale_fused.  Bit-identical to captra_sa_scale_fused at ~60 % of its
 * flops for the SA2 scales.  Instantiated for (cfeat,c1,c2,c3) = (320,128,128,256), (320,128,196,256); -2 otherwise. */
int captra_sa_scale_pre(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1,
                        const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1, const float *w2,
                        const float *b2, const float *w3, const float *b3, float *out, int out_ctotal, int co_off,
                        captra_stream_t stream);

/* The same scale on the pipelined kernel (csrc/sa_pipe.hip: persistent one-wave-per-SIMD workgroups, next tile's gather
 * under the current tile's MFMAs, deferred epilogues, weight sets two ahead, staged output rows): v1pm is the POINT-major
 * (B,N,c1) result of captra_pointwise_mlp_pm.  Same k-ascending chains: bit-identical to captra_sa_scale_pre /
 * captra_sa_scale_fused.  w1, w2, w3 PACKED (both images, see PACKED WEIGHTS): the kernel streams the fragment images of
 * layers 2 / 3.  -2 when the shape is not instantiated or m*k is not a multiple of 128 (take captra_sa_scale_pre). */
int captra_sa_scale_pre_pm(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1pm,
                           const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1, const float *w2,
                           const float *b2, const float *w3, const float *b3, float *out, int out_ctotal, int co_off,
                           captra_stream_t stream);
int captra_sa_scale_pre_pm_ex(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1pm,
                           const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1, const float *w2,
                           const float *b2, const float *w3, const float *b3, float *out, int out_ctotal, int co_off,
                           const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* captra_pointwise_mlp2: the same layer on the channel concat [x; x2] WITHOUT building it -- SA3's [xyz, feat] of
 * sample_and_group_all (pointnet_utils.py:171-188) and FP3's [points1, repeat(points2)] (pointnet_utils.py:265-270).  x (B,csplit,L),
 * x2 (B,cin - csplit,L), or (B,cin - csplit) with x2_bcast = 1 (one vector per cloud, read for every position).  Rows are consumed
 * in the concat's order: bit-identical to captra_pointwise_mlp on the concatenated tensor.  -2: outside the instantiated shapes
 * (cout <= 64) or the tensors lie more than 2^30 bytes apart -- the caller concatenates. */
int captra_pointwise_mlp2(int b, int cin, int csplit, int cout, long long l, const float *x, const float *x2, int x2_bcast,
                          const float *wt_packed, const float *bias_packed, int act, float *y, captra_stream_t stream);
int captra_pointwise_mlp2_ex(int b, int cin, int csplit, int cout, long long l, const float *x, const float *x2, int x2_bcast,
                          const float *wt_packed, const float *bias_packed, int act, float *y, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */
/* captra_pointwise_mlp with a POINT-major result y (B,l,cout) (cout % 4 == 0, y 16-byte aligned; -2 otherwise or when the
 * layer is outside the direct-operand kernel's 32-bit offset range). */
int captra_pointwise_mlp_pm(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                            const float *bias_packed, int act, float *y, captra_stream_t stream);
int captra_pointwise_mlp_pm_ex(int b, int cin, int cout, long long l, const float *x, const float *wt_packed,
                            const float *bias_packed, int act, float *y, const captra_launch_opts *opts,
        captra_stream_t stream);   /* the same with per-call options */

/* Three dense layers in one launch: y (B,c3,l) = act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3), x (B,c0,l); packed
 * weights (captra_pack_weights).  Replaces the FP1 shared MLP + conv1/bn1/ReLU tail of PointNet2Msg
 * (pointnet_utils.py:296-298, backbones.py:66-68) without the two intermediate (B,128,l) tensors.  Instantiated
 * for (c0,c1,c2,c3) = (134,128,128,128) and (131,128,128,128); returns -2 for any other shape (run the layers
 * with captra_pointwise_mlp then).  Bit-identical to three captra_pointwise_mlp calls. */
int captra_mlp_chain3(int b, int c0, int c1, int c2, int c3, long long l, const float *x, const float *w1,
                      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int act3,
                      float *y, captra_stream_t stream);

/* CoordNet's tail in one launch (networks.py:29-32, 44-46; backbones.py:66-68): x (B,c0,l) = FP1's concatenated input ->
 * FP1 shared MLP (2 layers) -> conv1+bn1+ReLU -> feat (128 channels, never stored) -> seg (B,seg_dim,l) = segmentation head
 * (one conv, raw logits) and nocs (B,nocs_dim,l) = NOCS head (conv+BN+ReLU, conv, nocs_act: CAPTRA_ACT_SIGMOID_M05 gives
 * sigmoid - 0.5).  w / bias: HOST arrays of the six layers' packed device pointers in that order (fp1a, fp1b, conv1, seg,
 * nocs hidden, nocs out); hidden widths 128.  Instantiated for c0 = 134 and (seg_dim, nocs_dim) in {(2,3), (4,12), (3,9), (2,6)};
 * -2 otherwise.  Every output equals the corresponding chain of captra_pointwise_mlp calls bit for bit (sigmoid: same expf). */
int captra_coord_tail(int b, int c0, int seg_dim, int nocs_dim, long long l, const float *x, const float *const *w,
                      const float *const *bias, int nocs_act, float *seg, float *nocs, captra_stream_t stream);

/* Feature propagation input (pointnet_utils.py:280-294, CUDA semantics: weights from sqrt(d2), SURVEY.md
 * §2.2), in two halves so that networks looking at the same cloud share the geometric one:
 *   captra_three_nn_weights: unknown (B,N,3), known (B,S,3) -> idx (B,N,3) i32, weight (B,N,3) f32 with
 *       w_j = (1/(sqrt(d2_j)+1e-8)) / sum_j(...);
 *   captra_interp_concat: skip (B,c1,N) or NULL, feat_known (B,c2,S), idx, weight ->
 *       out (B,c1+c2,N) = cat([skip, sum_j w_j feat_known[:, idx_j]]);
 *   captra_fp_interpolate_concat: both, with caller-provided scratch idx (B,N,3) / weight (B,N,3). */
int captra_three_nn_weights(int b, int n, int s, const float *unknown, const float *known, int *idx,
                            float *weight, captra_stream_t stream);
int captra_interp_concat(int b, int n, int s, int c1, int c2, const float *skip, const float *feat_known,
                         const int *idx, const float *weight, float *out, captra_stream_t stream);
int captra_fp_interpolate_concat(int b, int n, int s, int c1, int c2, const float *unknown,
                                 const float *known, const float *skip, const float *feat_known,
                                 int *idx_scratch, float *weight_scratch, float *out, captra_stream_t stream);

/* GroupNorm over groups of `channels_per_group` consecutive channels of x (B,C,N) + optional ReLU
 * (nn.GroupNorm(C/2, C) + ReLU of the rotation heads, blocks.py:70-71, 148-165): one pass, the
 * group's values stay in registers between the statistics and the normalisation.
 * N % 4 == 0 and channels_per_group * N <= 16384. */
int captra_group_norm_relu(int b, int c, int n, int channels_per_group, float eps, int relu, const float *x,
                           const float *gamma, const float *beta, float *y, captra_stream_t stream);

/* Masked Procrustes scale + translation fit for all (trajectory, part) pairs on device
 * (part_fit_st_no_ransac pose_fit.py:38-53 -> transform_pts_mask procrustes.py:132-164 with a
 * given rotation; sym adds the in-plane 2x2 SVD of procrustes.py:167-228).  No host round trip.
 *   labels (B,N) i32 (values >= P are background); src = predicted NOCS (B,P,3,N) channel-major;
 *   tgt = camera points, (B,3,N) shared by the parts (tgt_per_part = 0, what the track loop has)
 *   or (B,P,3,N) (tgt_per_part = 1); rot (B,P,3,3); given_scale (B,P) or NULL (pose_fit.py:38).
 *   -> scale (B,P), trans (B,P,3), valid (B,P) i32 = (count > 3) && finite(scale, trans, rot). */
int captra_part_fit_st(int b, int p, int n, int sym, const int *labels, const float *src,
                       const float *tgt, int tgt_per_part, const float *rot, const float *given_scale,
                       float *scale, float *trans, int *valid, captra_stream_t stream);

/* The track loop's form (networks.py:219-232 in one launch): target = pts (B,3,N) + pts_mean (B,3), formed in the kernel with the
 * same single fp32 addition as the reference's `points + points_mean`; a part whose fit is invalid keeps prev_scale (B,P) /
 * prev_trans (B,P,3) (NULL: the raw fit is written, as captra_part_fit_st does). */
int captra_part_fit_st_track(int b, int p, int n, int sym, const int *labels, const float *src, const float *pts,
                             const float *pts_mean, const float *rot, const float *prev_scale, const float *prev_trans,
                             float *scale, float *trans, int *valid, captra_stream_t stream);

/* CoordinateNet read-out (networks.py:50 F.softmax(dim=1) + model.py:466 torch.max(seg, dim=-2)[1]) in one launch: logits (B,S,N),
 * S <= 8 -> seg (B,S,N) softmax (or NULL), labels (B,N) i32 = FIRST index of the largest logit (or NULL). */
int captra_seg_softmax_argmax(int b, int s, int n, const float *logits, float *seg, int *labels, captra_stream_t stream);

/* njobs <= 16 device-to-device copies (bytes[j] % 4 == 0, 4-byte aligned) in one launch: host arrays of DEVICE pointers.  The
 * lanes' per-frame pose / record hand-over. */
int captra_copy_multi(int njobs, const void *const *src, void *const *dst, const long long *bytes, captra_stream_t stream);

/* out (rows) = max over each row of x (rows, l) fp32: the pooling of a group_all set abstraction (pointnet_utils.py:342,
 * torch.max(new_points, 2)[0]) on a (B,C,N) tensor, rows = B*C.  NaN handling as fmaxf (a NaN is ignored), unlike torch.max. */
int captra_row_max(long long rows, int l, const float *x, float *out, captra_stream_t stream);

/* The packed pose records of the per-frame exchange (no reference counterpart: the reference is one
 * process): n = B*P records [R(9) t(3) s(1) valid(1)] from rot (n,3,3), trans (n,3), scale (n), valid (n) floats (NULL = 1)
 * -> out0 (n,14) and, when non-NULL, a second copy out1 (the single-rank "gathered" buffer). */
int captra_pack_pose(int n, const float *rot, const float *trans, const float *scale, const float *valid, float *out0,
                     float *out1, captra_stream_t stream);

/* Rotation read-out of one tracking step (blocks.py:147-156, networks.py:127-138 and 200-203,
 * part_dof_utils.py:124-141, rotations.py:302-387) in one launch.  raw = the rotation heads' raw outputs on the B*P
 * canonicalised clouds (R = 3 symmetric: y-axis; R = 6: ortho6d): (B*P, P, R, N) with all P heads per cloud as the
 * reference evaluates them (diag_only = 0), or (B*P, R, N) holding only head p on cloud (b,p) -- the only entries the
 * read-out consumes (diag_only = 1).  labels (B,N) i32, prev_rot (B,P,3,3)
 *   -> rot (B,P,3,3) = prev_rot * dR, where dR = frame of the masked-mean per-point prediction of head p on cloud (b,p)
 *      ((0,1,0) / identity when no point carries label p); delta (B,P,3,3) receives dR when non-NULL. */
int captra_rot_pool_compose(int b, int p, int n, int sym, int diag_only, const float *raw, const int *labels,
                            const float *prev_rot, float *rot, float *delta, captra_stream_t stream);

/* Batched 3x3 orthogonal Procrustes: R = U diag(1,1,det(U V^T)) V^T with U S V^T = tgt^T src
 * (rotate_pts_batch procrustes.py:25-56).  src, tgt (nb,N,3) -> rot (nb,3,3).  One-sided Jacobi. */
int captra_procrustes_rot3(int nb, int n, const float *src, const float *tgt, float *rot,
                           captra_stream_t stream);

/* A set-abstraction LEVEL's scales in one call: njobs (<= 4) jobs, each the argument list of captra_sa_scale_fused (pre = 0:
 * feat_or_v1 = feat) or captra_sa_scale_pre_pm (pre = 1: feat_or_v1 = v1pm, b1 unused).  When the jobs are a level's three
 * small-input scales in order ([cf+3 -> 32 -> 32 -> 64], [-> 64 -> 64 -> 128], [-> 64 -> 96 -> 128]) -- and / or the second level's
 * two ([323 -> 128 -> 128 -> 256], [-> 128 -> 196 -> 256]) -- they run as ONE launch, each on its own range of workgroups (bits
 * unchanged: every workgroup does what its scale's own launch would have done); otherwise one launch after the other, exactly
 * as the per-scale calls.  For steps of few clouds, where a scale's own launch fills a fraction of the chip.  opts->dyn_slot is
 * ignored (the one-launch forms walk statically).  A pure function of its arguments: safe from any number of host threads. */
typedef struct captra_sa_scale_job {
    int pre, b, n, m, k, cfeat, c1, c2, c3;
    const float *feat_or_v1, *xyz_cn, *new_xyz;
    const int *idx;
    const float *w1, *b1, *w2, *b2, *w3, *b3;      /* packed buffers of the three layers */
    float *out;
    int out_ctotal, co_off;
} captra_sa_scale_job;
int captra_sa_scales_multi(int njobs, const captra_sa_scale_job *jobs, const captra_launch_opts *opts, captra_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Section 3 — introspection
 * ---------------------------------------------------------------------------------------- */
const char *captra_error_string(int err);
const char *captra_version(void);
/* Per-kernel HIP-event timing: when enabled every launcher brackets its kernel with events on
 * its own stream.  captra_prof_read synchronises the recorded events and returns accumulated
 * milliseconds / launch count for `name` (the kernel family, e.g. "ball_query"). */
void captra_prof_enable(int on);
void captra_prof_reset(void);
int captra_prof_read(const char *name, double *total_ms, long long *launches);
int captra_prof_names(char *buf, int buflen);

/* ---- 4. MEASUREMENT switches (NOT part of the stable ABI; used by tests/ and tools/ to cross-check or time variants that compute
 *         the same bits).  Nothing the product path needs is set here: what a caller wants different from the defaults travels with
 *         the call (captra_launch_opts).  Each switch is THREAD-LOCAL: it affects launches made by the calling host thread only.
 *         Defaults (0; 1 for captra_pw_set_direct / captra_pw_set_pair) select the production kernels. ---- */
void captra_fps_set_waves(int waves);       /* FPS: waves per cloud (0 = heuristic) */
void captra_fps_set_pruned_min(int n);     /* FPS: clouds of >= n points take the pruned kernel (default 8192; 0 = never) */
void captra_fps_set_stats(unsigned long long *dev_counters); /* pruned FPS: accumulate 6 counters {bucket updates, refreshes, cycles of 4 phases} (NULL = off) */
void captra_fps_set_variant(int v);         /* FPS: 0 = blocked ownership + ballot pick (default), 1 = first-generation kernel */
void captra_sa_fused_set_mode(int mode);    /* SA scale: 0 = register-resident kernels where instantiated, 1 = generic LDS kernel
                                               for every shape, 2 = register-resident with streamed weights only */
void captra_sa_fused_set_wn(int wn);        /* generic LDS kernel: sub-tile width 32*wn (0 = heuristic) */
void captra_sa_fused_set_prof(unsigned long long *dev_counters); /* sa_wave_kernel: 10 device counters of phase timers, or NULL */
void captra_pw_set_direct(int on);          /* dense layers: 1 = direct-operand kernel (default), 0 = LDS-staged kernel */
void captra_ball_query_set_prune(int on);   /* ball query: 1 = small radii of 1024..4096-point clouds from a cell grid (exact; measured slower, off by default), 0 = index-order scan */
void captra_pw_set_occupancy(int occ);      /* dense layers, 64x64 wave tiles: workgroups per CU, 0 / 4 = as built (default), 3 / 2 = fewer (measurements) */
void captra_pw_set_dbg(int v);             /* dense layers of a GroupNorm chain: timing ablations (-DCAPTRA_ABLATIONS=1 builds only, results wrong; ignored otherwise) */
void captra_pw_set_pair(int on);            /* dense layers: 1 = paired column tiles where L is even (default), 0 = never */
void captra_sa1_stream_set_grid(int grid, int prio); /* level-1 stream kernel: workgroups (0 = two per CU), 1 = samplers at s_setprio 3 (default) */
void captra_sa1_stream_set_fine(int centres); /* level-1 stream kernel: trailing centres handed out as fine tickets of 8 (multiple of 32, default 32) */
void captra_sa1_stream_set_whole(int windows); /* level-1 stream kernel: bits 0-7 = leading windows of 32 centres handed out as one ticket for all three scales (default 0), bit 8 = three scale tickets per window instead of two */
void captra_sa_bf16_set_variant(int v);     /* bf16 SA scales: bit 0 = small-input scales without gather prefetch / fragment ring, bit 3 = SA2 scales on
                                               sa_bf16_kernel; bits 4.. (ablations with WRONG results, timing only) exist in -DCAPTRA_ABLATIONS=1 builds only */
void captra_neck_chain_set_split(int n); /* captra_neck_chain_bf16, three-layer modules: workgroups sharing the last layer of a position tile (1 / 2 / 4, default 4) */
void captra_query_and_group_set_shape(int mcb, int cc); /* captra_query_and_group: centres per workgroup, channels per chunk (0 = heuristic) */
void captra_group_set_shape(int lds_kb, int ccmax, int ppb); /* group_points: staging budget (KiB, <= 64), channels per workgroup, positions per
                                               workgroup (0 = default); tools/bench_group.py --sweep */

#ifdef __cplusplus
}
#endif
#endif /* CAPTRA_HIP_H */
