// sa_wave_pipe_kernel<CF, C1, C2, C3>: the register-resident SA2 scale (sa_fused.hip: sa_wave_kernel<..., PRE>) rebuilt
// around ONE wave per SIMD and its whole 512-register file, for gfx950.
//
// What the counters said about sa_wave_kernel<320,128,196,256,true> (profiles/r02a_*; tools/bench_sa_fused.py --phases):
// it already ran at one wave per SIMD (198 VGPR + 64 AGPR under the latency-first scheduler), so nothing covered
//   * the first layer's gather (64 four-byte gathers per lane from the channel-major v1, up to 64 cache lines per
//     instruction): 15 % of a wave's life,
//   * the epilogues between output-tile passes (ReLU + permlane swap; ReLU + five DPP max steps + LDS): ~9 %,
//   * late weight sets (one 16-register set = 1024 MFMA cycles of lead against an L2 hit under load): ~7 %.
// Here the same k-ascending fmaf chains (bit-identical results) are scheduled so that the matrix pipe always has work:
//   1. a WAVE owns a centre and walks its K neighbours in K / 32 slices (static walk over the centres); the NEXT slice's
//      neighbour ids, centre and accumulator start values are fetched while the current slice's layers 2 / 3 run (the
//      gather costs registers, not time).  v1 is read POINT-major (B,N,C1; written that way by captra_pointwise_mlp_pm):
//      the four accumulator rows a lane needs are one 16-byte load, a slice touches 128 cache lines instead of up to 4096;
//   2. every pass's epilogue is deferred by one pass and issued in parts behind the next pass's MFMA blocks (two
//      accumulator pairs alternate), only a layer's last pass is exposed;
//   3. weight sets stream two sets (2048 MFMA cycles) ahead through a ring of three;
//   4. the last layer's maximum over the centre's slices stays in registers (a transpose-reduce butterfly per slice folded
//      into a running maximum, sw_bfly_finish once per centre): no LDS, no barrier after start-up.  (Round 2 first staged
//      the maxima of chunks of 8 tiles in LDS and wrote consecutive centres per row -- one barrier per chunk -- and handed
//      chunks out through an atomic ticket counter; the wave-per-centre form is 1-2 % faster in the step and needs
//      neither.  The 4-byte stores strided along M it brings back are merged in L2: WRITE_SIZE stays near the output size.)
// Replaces nothing of the reference one-to-one: it is the body of PointNetSetAbstractionMsg.forward's loop over radii
// (pointnet_utils.py:228-248) for the SA2 shapes, as sa_wave_kernel is.
#include "wave_mlp.h"

namespace {

constexpr int SP_POS = 128;   // the launcher's tiling unit (M*K must be a multiple of it: shapes of the backbone)

struct SpParams {
    int b, n, m, k;
    const float *v1pm;      // (B,N,C1) point-major: b1 + W1[feature rows] feat per source point
    const float *xyz_cn;    // (B,3,N)
    const float *new_xyz;   // (B,M,3)
    const int *idx;         // (B,M,K)
    const float *w1;        // packed first-layer weights (rows CF..CF+2 = the relative-xyz rows are used)
    const float *w2, *b2, *w3, *b3;   // packed (row-major + fragment images); layers 2 / 3 stream the fragment image
    float *out;             // (B,out_ctotal,M)
    int out_ctotal, co_off;
    unsigned long long *prof;  // debug: per-phase wave-cycle totals of a sample of workgroups (captra_sa_fused_set_prof), or null
    int split;                 // a wave owns ONE 32-neighbour slice of a centre (small batches); maxima combined by atomic max on the zeroed output
    int *dyn;                  // zeroed counter: centres beyond a wave's first are handed out through it (sa_fused.hip, captra_sa_set_dynamic)
};

#define SP_TICK(slot)                                                           \
    if (CAPTRA_PROF_ON(p.prof)) {                                                    \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();          \
        if (lane == 0 && sampled) atomicAdd(p.prof + (slot), t_now - t_last);   \
        t_last = t_now;                                                         \
    }

// ---- deferred epilogues, one part at a time -------------------------------------------------------------------------
// MID: unit u = (tile tm = u >> 2, register quad q = u & 3); MAX: unit u = (tile tm = u >> 4, register r = u & 15).
// Part c of NPARTS takes the units with u * NPARTS / UNITS == c, so the work spreads evenly over a pass's k-step sets.
template <int NOUT>
__device__ __forceinline__ void sp_mid_unit(const f32x16 &acc, int t, int q, float (&hout)[NOUT]) {
    unsigned a[4];          // ReLU on the bit pattern: one v_max_i32 (the float form is canonicalise + max)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = __float_as_int(acc[4 * q + i]);
        a[i] = (unsigned)(x > 0 ? x : 0);
    }
    const auto p01 = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);
    const auto p23 = __builtin_amdgcn_permlane32_swap(a[2], a[3], false, false);
    const int k0 = 16 * t + 4 * q;
    if (k0 + 0 < NOUT) hout[k0 + 0] = __uint_as_float(p01[0]);
    if (k0 + 1 < NOUT) hout[k0 + 1] = __uint_as_float(p23[0]);
    if (k0 + 2 < NOUT) hout[k0 + 2] = __uint_as_float(p01[1]);
    if (k0 + 3 < NOUT) hout[k0 + 3] = __uint_as_float(p23[1]);
}

// ---- max over the 32 positions of a tile: a transpose-reduce butterfly ---------------------------------------------------
// The old epilogue reduced each of a tile's 16 accumulator registers on its own (ReLU + five DPP max steps + a one-lane LDS
// store under an exec mask: ~8 instructions and an exec switch per register, ~130 per tile).  Here neighbouring REGISTERS
// are merged while neighbouring LANES are reduced: stage k pairs lane l with lane l ^ 2^k and registers (2i, 2i+1); a lane
// with bit k clear keeps register 2i and takes the partner's 2i, a lane with bit k set keeps 2i+1.  After four stages ONE
// register holds, in lane l, the maximum over its row of 16 lanes of accumulator register (l & 15): 16 + 8 + 4 + 2 max
// steps instead of 80, one ReLU instead of 16 (max on the raw bit patterns: if any input is positive the signed-integer
// maximum is the largest positive float, otherwise the result is negative and the ReLU makes it 0 -- what ReLU-then-max
// gives), and ONE 64-lane ds_write_b32 per tile.  The two rows of 16 of each half-wave are combined by the final cross-wave
// pass, which reads 2 x 4 slots per output row instead of 4.  51 VALU + 1 DS per tile.
__device__ __forceinline__ int sp_imax(int a, int b) { return a > b ? a : b; }

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ int sp_dpp(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK_MASK, false);
}

struct SpMaxState {
    int w[8];   // after stage A (lane bit 0)
    int u[2];   // after stages B, C (lane bits 1, 2)
};

// stage A: 16 raw accumulator registers -> 8
__device__ __forceinline__ void sp_max_stage_a(const f32x16 &acc, SpMaxState &st, int lane) {
    const bool b0 = lane & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int x0 = __float_as_int(acc[2 * i]), x1 = __float_as_int(acc[2 * i + 1]);
        const int own = b0 ? x1 : x0, oth = b0 ? x0 : x1;
        st.w[i] = sw_imax_dpp<0xB1>(own, oth);                            // quad_perm [1,0,3,2]: lane ^ 1
    }
}

// stages B and C: 8 -> 4 -> 2
__device__ __forceinline__ void sp_max_stage_bc(SpMaxState &st, int lane) {
    const bool b1 = lane & 2, b2 = lane & 4;
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int own = b1 ? st.w[2 * j + 1] : st.w[2 * j], oth = b1 ? st.w[2 * j] : st.w[2 * j + 1];
        v[j] = sw_imax_dpp<0x4E>(own, oth);                                // quad_perm [2,3,0,1]: lane ^ 2
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int own = b2 ? v[2 * m + 1] : v[2 * m], oth = b2 ? v[2 * m] : v[2 * m + 1];
        // lane ^ 4: banks 0, 2 (lanes 0-3, 8-11 of a row) read four lanes up, banks 1, 3 four lanes down
        const int p = sp_dpp<0x104, 0x5>(own, oth);                      // row_shl:4 into banks 0, 2; the others keep `own`
        const int q = sp_dpp<0x114, 0xA>(p, oth);                        // row_shr:4 into banks 1, 3
        st.u[m] = sp_imax(own, q);
    }
}

// stage D, folded into the centre's running maximum: zr's lane l holds, for accumulator register (l & 15) = output row
// 32 t + 8 (r >> 2) + (r & 3) + 4 (l >> 5), the maximum over the row of 16 lanes and over every slice so far (zr starts at
// 0 = the ReLU; the two rows of 16 of a half-wave are joined once per centre, sw_bfly_finish)
__device__ __forceinline__ void sp_max_stage_d(const SpMaxState &st, int &zr) {
    const int lane = (int)(threadIdx.x & 63);
    const bool b3 = lane & 8;
    const int own = b3 ? st.u[1] : st.u[0], oth = b3 ? st.u[0] : st.u[1];
    zr = sw_imax(zr, sw_imax_dpp<0x128>(own, oth));                        // row_ror:8: lane ^ 8 within the row
}

template <int COUT, bool LAST, int NPARTS, int NOUT, int NZ>
__device__ __forceinline__ void sp_epi_part(const f32x16 (&acc)[2], int ps, int c, float (&hout)[NOUT], int (&zrun)[NZ],
                                            SpMaxState (&mst)[2], int lane) {
    constexpr int NT = (COUT + 31) / 32;
    if constexpr (LAST) {
        // six units per pass: (tile 0: A, BC, D) (tile 1: A, BC, D), spread over the NPARTS steps of the next pass
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (u * NPARTS / 6 == c && 2 * ps + u / 3 < NT) {
                const int tm = u / 3;
                if (u % 3 == 0) sp_max_stage_a(acc[tm], mst[tm], lane);
                else if (u % 3 == 1) sp_max_stage_bc(mst[tm], lane);
                else sp_max_stage_d(mst[tm], zrun[2 * ps + tm < NZ ? 2 * ps + tm : 0]);
            }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u * NPARTS / 8 == c && 2 * ps + (u >> 2) < NT) sp_mid_unit<NOUT>(acc[u >> 2], 2 * ps + (u >> 2), u & 3, hout);
    }
}

// Weight sets: the fragment image of the packed buffer (wave_mlp.h: sw_load_set / sw_first_set), 4 loads per set instead of
// 16.  With one dword load per MFMA the four waves of a CU kept its vector-memory address unit (one 64-lane instruction per
// ~16 cycles) exactly as busy as its matrix pipes (one MFMA per SIMD per 64 cycles): 19 % of a tile's cycles went into
// waiting for weights that were in L2 all along (measured in round 2 with the weight loads compiled out).

// One layer: hin[] (B operands) -> hout[] (LAST = false) or per-wave maxima in red (LAST = true).  Weight sets in the ring
// s[3]: the set of step g is s[(START + g) % 3]; this layer's first set must already be on its way (previous phase), the
// following layer's first set is requested through `next(s[(START + STEPS) % 3])` two steps before the end.  `side(g)` runs
// once per step before the MFMA block (the kernel hangs the next tile's prefetch on it).
template <int CIN, int COUT, bool LAST, int START, int NIN, int NOUT, int NZ, typename Next, typename Side>
__device__ __forceinline__ void sp_layer(const float *wt, const float *bias_lds, const float (&hin)[NIN], float (&hout)[NOUT],
                                         float (&s)[3][16], int (&zrun)[NZ], int lane, Next next, Side side) {
    using S = SwShape<CIN, COUT>;
    static_assert(NIN >= S::KST, "input operand array too small");
    const __amdgpu_buffer_rsrc_t rsrc = sw_frag_rsrc<CIN, COUT>(wt);
    const int voff = lane * 16;
    f32x16 acc[2][2];
    SpMaxState mst[2];
    static_assert(!LAST || S::NSETS >= 6 || S::NPASS == 1, "the deferred max needs its six units in distinct steps, in order");
#pragma unroll
    for (int g = 0; g < S::STEPS; ++g) {
        const int ps = g / S::NSETS, c = g % S::NSETS, pb = ps & 1;
        if (c == 0) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                if (2 * ps + tm < S::NT) sw_bias_init(acc[pb][tm], bias_lds, 2 * ps + tm, lane);
        }
        if (g == 0 && S::STEPS > 1) sw_load_set<CIN, COUT>(s[(START + 1) % 3], rsrc, voff, 1 / S::NSETS, 1 % S::NSETS);
        if (g + 2 < S::STEPS) sw_load_set<CIN, COUT>(s[(START + g + 2) % 3], rsrc, voff, (g + 2) / S::NSETS, (g + 2) % S::NSETS);
        if (g + 2 == S::STEPS || (S::STEPS == 1 && g == 0)) next(s[(START + S::STEPS) % 3]);
        side(g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < SW_KS; ++j)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const int kk = c * SW_KS + j;
                if (kk < S::KST && 2 * ps + tm < S::NT)
                    acc[pb][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[(START + g) % 3][tm * SW_KS + j], hin[kk], acc[pb][tm], 0, 0, 0);
            }
        // the previous pass's epilogue, one part per step: issued behind this step's MFMAs, it runs while they execute
        if (ps > 0) {
            sp_epi_part<COUT, LAST, S::NSETS>(acc[pb ^ 1], ps - 1, c, hout, zrun, mst, lane);
            // one MFMA, then a few of the part's VALU instructions, sixteen times: each of them issues while an MFMA executes
            // (left to itself the scheduler puts the whole part behind the block, where only the last MFMA covers it)
#pragma unroll
            for (int i = 0; i < 2 * SW_KS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int LP = S::NPASS - 1;
#pragma unroll
    for (int c = 0; c < S::NSETS; ++c) sp_epi_part<COUT, LAST, S::NSETS>(acc[LP & 1], LP, c, hout, zrun, mst, lane);
}

// (the kernel's body as a device function of (workgroup, workgroups): sa_wave_pipe2_kernel runs the level's two scales in one launch)
template <int CF, int C1, int C2, int C3>
__device__ __forceinline__ void sp_body(const SpParams &p, const int blk, const int nblk) {
    using S1 = SwShape<CF + 3, C1>;
    using S2 = SwShape<C1, C2>;
    using S3 = SwShape<C2, C3>;
    constexpr int NT1 = S1::NT;
    static_assert(NT1 <= 4 && C1 % 32 == 0, "first-layer width");
    __shared__ __attribute__((aligned(16))) float bias_lds[2 * 256];       // packed b2, b3 (zero padded)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int e = tid; e < 2 * 256; e += 256) {
        const int c = e & 255;
        bias_lds[e] = e < 256 ? (c < pad128c(C2) ? p.b2[c] : 0.f) : (c < pad128c(C3) ? p.b3[c] : 0.f);
    }
    // first layer's xyz rows of W1 (A operands of its two k-steps): the same for every slice
    float at[2][NT1];
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, S1::KP * S1::LDW * 4, 0x00020000);
        const int voff_w = (half * S1::LDW + (lane & 31)) * 4;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int t = 0; t < NT1; ++t)
                at[jj][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, voff_w, ((CF + 2 * jj) * S1::LDW + 32 * t) * 4, 0));
    }
    __syncthreads();

    // A WAVE owns a centre (as in sa_wave_lds_kernel): it walks the centre's K neighbours in K / 32 slices and keeps the
    // last layer's running maximum in registers, so the four waves of a workgroup share nothing but the biases: no LDS
    // staging of maxima, no chunk barrier, no read-out pass.  Centres are walked statically: gid, gid + nwaves, ...
    const int nwaves = nblk * 4, gid = blk * 4 + wave;
    const int ncentres = p.b * p.m;                       // < 2^30 (launcher)
    const int nslices = p.k / 32;
    // a slice's per-lane inputs: neighbour id, relative xyz operands, accumulator start values (gathered v1 rows)
    int id = 0;
    float bt[2] = {0.f, 0.f};
    float4 g4[NT1][4];
    auto load_id = [&](int c, int sl) { return p.idx[(size_t)c * p.k + sl * 32 + (lane & 31)]; };
    auto load_rest = [&](int c, int id_, float (&bt_)[2], float4 (&g_)[NT1][4]) {
        const int tb = c / p.m;
        const float *cp = p.new_xyz + (size_t)c * 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int a = 2 * jj + half;
            bt_[jj] = a < 3 ? p.xyz_cn[((size_t)tb * 3 + a) * p.n + id_] - cp[a] : 0.f;
        }
        const float4 *vp = reinterpret_cast<const float4 *>(p.v1pm + ((size_t)tb * p.n + id_) * C1 + 4 * half);
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) g_[t][q] = vp[8 * t + 2 * q];       // rows 32t + 8q + 4 half + (0..3)
    };

    const bool sampled = blk % 16 == 0;
    unsigned long long t_last = CAPTRA_PROF_ON(p.prof) ? __builtin_amdgcn_s_memtime() : 0ull;
    float s[3][16];
    constexpr int START3 = S2::STEPS % 3;                     // ring slot of layer 3's first set (layer 2 starts in slot 0)
    constexpr int NEXT2 = (START3 + S3::STEPS) % 3;            // slot in which layer 3 leaves the NEXT slice's first layer-2 set
    int zrun[S3::NT];
    // SPLIT (small batches, as in sa_wave_lds_kernel): a wave owns ONE slice -- task t = gid, gid + nwaves, ... is slice
    // t % nslices of centre t / nslices -- and a centre's slices meet in an integer atomic max on the pre-zeroed output
    // (non-negative floats order like their bit patterns; max is exact and order-free: the running maximum's bits).
    const bool split = p.split != 0;
    const int ssh = __builtin_ctz((unsigned)nslices);
    int task = gid;
    int c = split ? gid >> ssh : gid, sl = split ? gid & (nslices - 1) : 0;
    // dynamic hand-out of the centres after a wave's first (as in sa_wave_lds_kernel): asked for at a centre's first slice, read
    // at its last
    const bool dyn = p.dyn != nullptr && !split && nslices > 1;
    int dyn_raw = 0, c_next = gid + nwaves;
    if (dyn) {
        if (lane == 0) dyn_raw = atomicAdd(p.dyn, 1);
        c = __builtin_amdgcn_readfirstlane(dyn_raw);
    }
    if (c < ncentres) {
        sw_first_set<C1, C2>(s[NEXT2], p.w2, lane);
        id = load_id(c, sl);
        load_rest(c, id, bt, g4);
    }
    while (c < ncentres) {
        if (sl == 0 || split) {
#pragma unroll
            for (int t = 0; t < S3::NT; ++t) zrun[t] = 0;          // (0 = the ReLU)
        }
        // ---- what comes after this slice (wave-uniform) ----
        const bool last_slice = split || sl + 1 == nslices;
        if (dyn) {
            if (sl == 0 && lane == 0) dyn_raw = atomicAdd(p.dyn, 1);
            if (last_slice) c_next = __builtin_amdgcn_readfirstlane(dyn_raw);
        } else {
            c_next = c + nwaves;
        }
        const int cn = split ? (task + nwaves) >> ssh : (last_slice ? c_next : c), sn = split ? (task + nwaves) & (nslices - 1) : (last_slice ? 0 : sl + 1);
        const bool has_next = cn < ncentres;
        int id_n = 0;

        float h1[S2::KST], h2[S3::KST], none[1];
        SP_TICK(0)
        // layer 2's first weight set was requested two steps before the previous slice ended: move it to slot 0
        if (NEXT2 != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[0][i] = s[NEXT2][i];
        }
        // ---- layer 1: the chain continues from the gathered start values with the two relative-xyz k-steps ----
        {
            f32x16 acc[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[t][4 * q + 0] = g4[t][q].x; acc[t][4 * q + 1] = g4[t][q].y;
                    acc[t][4 * q + 2] = g4[t][q].z; acc[t][4 * q + 3] = g4[t][q].w;
                }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int t = 0; t < NT1; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[jj][t], bt[jj], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT1; ++t) sw_mid_epilogue<S2::KST>(acc[t], t, h1);
        }
        __builtin_amdgcn_sched_barrier(0);
        SP_TICK(1)
        // ---- layer 2; its second step asks for the next slice's neighbour ids ----
        sp_layer<C1, C2, false, 0>(p.w2, bias_lds, h1, h2, s, zrun, lane,
                                   [&](float (&dst)[16]) { sw_first_set<C2, C3>(dst, p.w3, lane); },
                                   [&](int g) { if (g == 1 && has_next) id_n = load_id(cn, sn); });
        SP_TICK(2)
        // ---- layer 3 + running max over the slices; its third step gathers the next slice's start values ----
        sp_layer<C2, C3, true, START3>(p.w3, bias_lds + 256, h2, none, s, zrun, lane,
                                       [&](float (&dst)[16]) { sw_first_set<C1, C2>(dst, p.w2, lane); },
                                       [&](int g) { if (g == 2 && has_next) load_rest(cn, id_n, bt, g4); });
        id = id_n;
        SP_TICK(3)
        if (CAPTRA_PROF_ON(p.prof) && lane == 0 && sampled) atomicAdd(p.prof + 9, 1ull);
        if (last_slice) {
            // the centre's maxima: lane l with (l & 16) == 0 holds row 32 t + 8 ((l & 15) >> 2) + (l & 3) + 4 (l >> 5) of tile t
            const int tb = c / p.m, centre = c - tb * p.m;
            const int r = lane & 15;
            const int row0 = 8 * (r >> 2) + (r & 3) + 4 * half;
            float *op = p.out + ((size_t)tb * p.out_ctotal + p.co_off + row0) * p.m + centre;
#pragma unroll
            for (int t = 0; t < S3::NT; ++t) {
                const float v = sw_bfly_finish(zrun[t]);
                if ((lane & 16) == 0 && 32 * t + row0 < C3) {
                    if (split) atomicMax(reinterpret_cast<int *>(op + (size_t)32 * t * p.m), __float_as_int(v));
                    else op[(size_t)32 * t * p.m] = v;
                }
            }
            SP_TICK(4)
        }
        c = cn; sl = sn; task += nwaves;
    }
}


template <int CF, int C1, int C2, int C3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sa_wave_pipe_kernel(SpParams p) {
    sp_body<CF, C1, C2, C3>(p, (int)blockIdx.x, (int)gridDim.x);
}

// The second level's two scales in one launch (few clouds: 64 + 128 workgroups of a chip that holds 256, one after the other
// otherwise): workgroups [0, g0) run the first, the rest the second -- each exactly what its own launch would have run.
struct Sp2Params { SpParams s[2]; int g0; };
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sa_wave_pipe2_kernel(Sp2Params q) {
    const int blk = (int)blockIdx.x;
    if (blk < q.g0) sp_body<320, 128, 128, 256>(q.s[0], blk, q.g0);
    else sp_body<320, 128, 196, 256>(q.s[1], blk - q.g0, (int)gridDim.x - q.g0);
}

}  // namespace

extern unsigned long long *captra_sa_prof_ptr();   // sa_fused.hip: the debug counters set by captra_sa_fused_set_prof
extern int captra_sa_split_knob();                  // sa_fused.hip: captra_sa_fused_set_split
extern int *captra_sa_dyn_slot(const captra_launch_opts *o, hipStream_t stream);   // sa_fused.hip: captra_launch_opts::dyn_slot

// the level's scales recorded by captra_sa_scales_multi (sa_fused.hip) into ITS collector's buffer, launched by the flush
struct SpRecord { SpParams q; unsigned grid; int code; };
struct SpCollect { SpRecord rec[2]; int n; };
void captra_sp_collect_init(void *buf, size_t bytes) {
    static_assert(sizeof(SpCollect) <= 1024, "SlCollect::sp (sa_fused.hip) holds an SpCollect");
    if (buf != nullptr && bytes >= sizeof(SpCollect)) static_cast<SpCollect *>(buf)->n = 0;
}
int captra_sp_collect_flush(void *buf, hipStream_t s) {
    SpCollect &c = *static_cast<SpCollect *>(buf);
    const int n = c.n;
    c.n = 0;
    if (n == 2 && c.rec[0].code == 0 && c.rec[1].code == 1 && c.rec[0].q.dyn == nullptr && c.rec[1].q.dyn == nullptr) {
        Sp2Params q2;
        q2.s[0] = c.rec[0].q; q2.s[1] = c.rec[1].q; q2.g0 = (int)c.rec[0].grid;
        CAPTRA_LAUNCH("sa_scale_fused", sa_wave_pipe2_kernel, dim3(c.rec[0].grid + c.rec[1].grid), dim3(256), 0, s, q2);
        return captra_last_error();
    }
    for (int i = 0; i < n; ++i) {
        if (c.rec[i].code == 0) { CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_pipe_kernel<320, 128, 128, 256>), dim3(c.rec[i].grid), dim3(256), 0, s, c.rec[i].q); }
        else { CAPTRA_LAUNCH("sa_scale_fused", (sa_wave_pipe_kernel<320, 128, 196, 256>), dim3(c.rec[i].grid), dim3(256), 0, s, c.rec[i].q); }
    }
    return captra_last_error();
}

// SA scale with a pre-transformed, POINT-major first layer (see include/captra_hip.h): v1pm (B,N,c1).
// -2: shape not instantiated / not tileable (the caller takes captra_sa_scale_pre).
int captra_sa_scale_pre_pm_impl(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1pm,
                                const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                                const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                                int out_ctotal, int co_off, const captra_launch_opts *opts, void *spbuf, captra_stream_t stream) {
    SpCollect *col = static_cast<SpCollect *>(spbuf);
    if (b < 0 || n < 1 || m < 0 || k < 1 || cfeat < 1 || c1 < 1 || c2 < 1 || c3 < 1 || v1pm == nullptr) return -1;
    if (out_ctotal < co_off + c3 || co_off < 0) return -1;
    if (k % 32 != 0 || 128 % k != 0) return -2;
    const long long L = (long long)m * k;
    if (L % SP_POS != 0 || (long long)c1 * n * 4 >= (1ll << 31) || (long long)b * L >= (1ll << 31)) return -2;
    if (b == 0 || m == 0) return 0;
    SpParams q;
    q.b = b; q.n = n; q.m = m; q.k = k; q.v1pm = v1pm; q.xyz_cn = xyz_cn; q.new_xyz = new_xyz; q.idx = idx;
    q.w1 = w1; q.w2 = w2; q.b2 = b2; q.w3 = w3; q.b3 = b3; q.out = out; q.out_ctotal = out_ctotal; q.co_off = co_off;
    q.prof = captra_sa_prof_ptr();
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    static std::atomic<int> cus_of[128];
    cus = cus_of[dev & 127].load(std::memory_order_relaxed);
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_of[dev & 127].store(cus, std::memory_order_relaxed);
    }
    cus = cus - captra_reserved_cus(opts) > 0 ? cus - captra_reserved_cus(opts) : 1;     // (captra_launch_opts::reserved_cus)
    // one workgroup (4 waves, one per SIMD) per CU; a wave per centre
    const long long centres = (long long)b * m;
    if (centres >= (1ll << 30)) return -2;
    const int split_knob = captra_sa_split_knob();
    q.split = (split_knob != 0 && k > 32 && (split_knob == 2 || centres < 4ll * cus)) ? 1 : 0;     // fewer centres than resident waves
    const long long wgs = ((q.split ? centres * (k / 32) : centres) + 3) / 4;
    const unsigned grid = (unsigned)(wgs < cus ? wgs : cus);
    q.dyn = (!q.split && k > 32 && wgs > 1 && col == nullptr) ? captra_sa_dyn_slot(opts, (hipStream_t)stream) : nullptr;
    if (q.split && !(opts != nullptr && opts->sa_prezeroed)) {
        // (zeroed by a kernel per cloud where the rows allow it, not by a memset node: common.h captra_zero_async)
        if (b <= 8 && ((size_t)c3 * m * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(out + (size_t)co_off * m) & 15) == 0 && ((size_t)out_ctotal * m * 4) % 16 == 0) {
            for (int bb = 0; bb < b; ++bb) (void)captra_zero_async(out + ((size_t)bb * out_ctotal + co_off) * m, (size_t)c3 * m * 4, (hipStream_t)stream);
        } else {
            (void)hipMemset2DAsync(out + (size_t)co_off * m, (size_t)out_ctotal * m * 4, 0, (size_t)c3 * m * 4, b, (hipStream_t)stream);
        }
    }
#define SPP_CASE(CF_, C1_, C2_, C3_)                                                                                  \
    if (cfeat == CF_ && c1 == C1_ && c2 == C2_ && c3 == C3_) {                                                        \
        auto kern = sa_wave_pipe_kernel<CF_, C1_, C2_, C3_>;                                                          \
        if (col != nullptr && col->n < 2) {                                                                           \
            col->rec[col->n].q = q; col->rec[col->n].grid = grid; col->rec[col->n].code = (C2_ == 128 ? 0 : 1);       \
            ++col->n;                                                                                                 \
            return 0;                                                                                                 \
        }                                                                                                             \
        CAPTRA_LAUNCH("sa_scale_fused", kern, dim3(grid), dim3(256), 0, (hipStream_t)stream, q);                       \
        return captra_last_error();                                                                                   \
    }
    SPP_CASE(320, 128, 128, 256)
    SPP_CASE(320, 128, 196, 256)
#undef SPP_CASE
    return -2;
}

extern "C" int captra_sa_scale_pre_pm_ex(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1pm,
                                         const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                                         const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                                         int out_ctotal, int co_off, const captra_launch_opts *opts, captra_stream_t stream) {
    return captra_sa_scale_pre_pm_impl(b, n, m, k, cfeat, c1, c2, c3, v1pm, xyz_cn, new_xyz, idx, w1, w2, b2, w3, b3, out, out_ctotal, co_off, opts, nullptr, stream);
}
extern "C" int captra_sa_scale_pre_pm(int b, int n, int m, int k, int cfeat, int c1, int c2, int c3, const float *v1pm,
                                      const float *xyz_cn, const float *new_xyz, const int *idx, const float *w1,
                                      const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                                      int out_ctotal, int co_off, captra_stream_t stream) {
    return captra_sa_scale_pre_pm_impl(b, n, m, k, cfeat, c1, c2, c3, v1pm, xyz_cn, new_xyz, idx, w1, w2, b2, w3, b3, out, out_ctotal, co_off, nullptr, nullptr, stream);
}
