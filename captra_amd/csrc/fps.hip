// Furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (reference sampling_gpu.cu:93-209): one workgroup per
// cloud, M-1 dependent selection rounds.  The reference keeps the running min-distance array in
// global memory and reduces through a 1+log2(block) barrier tree in shared memory per round.
// Here the cloud and its running min distances live in REGISTERS (PPT points per lane), a round
// is: broadcast-read the last pick from LDS -> PPT distance updates -> two 6-step DPP wave
// reductions (max distance, then min index among the maxima = lowest-index tie-break) -> ONE
// barrier with ping-pong LDS slots -> 4-step DPP reduction across the <=16 waves.
//
// Contract: idx[0] = 0; distance ((dx*dx+dy*dy)+dz*dz) in unfused fp32;
// running min = min(d, temp[k]); argmax with strict '>' scanning k ascending, i.e. the lowest
// index among equal maxima.  Distances are >= 0, so their IEEE bit patterns order like unsigned
// integers and the reductions run on u32.
#include "common.h"
#include "fps_round.h"

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    // lanes whose source is invalid / masked keep their own value (old = v)
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// reductions over a row of 16 lanes: afterwards every lane of the row holds the row result
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
    v = max(v, dpp_mov<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_mov<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_mov<0x141, 0xF>(v));  // row_half_mirror
    v = max(v, dpp_mov<0x140, 0xF>(v));  // row_mirror
    return v;
}
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
    v = min(v, dpp_mov<0xB1, 0xF>(v));
    v = min(v, dpp_mov<0x4E, 0xF>(v));
    v = min(v, dpp_mov<0x141, 0xF>(v));
    v = min(v, dpp_mov<0x140, 0xF>(v));
    return v;
}
// full wave64 reductions, result returned wave-uniform (SGPR)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = row_max_u32(v);
    v = max(v, dpp_mov<0x142, 0xA>(v));  // row_bcast15 -> rows 1,3
    v = max(v, dpp_mov<0x143, 0xC>(v));  // row_bcast31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = row_min_u32(v);
    v = min(v, dpp_mov<0x142, 0xA>(v));
    v = min(v, dpp_mov<0x143, 0xC>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// NWAVES: waves per workgroup (power of two <= 16); PPT: points held per lane;
// XYZ_LDS: whole cloud mirrored in LDS (SoA) for the broadcast read of the last pick.
template <int NWAVES, int PPT, bool XYZ_LDS>
__global__ __launch_bounds__(NWAVES * 64) void fps_kernel(int n, int m,
                                                          const float *__restrict__ xyz_all,
                                                          float *__restrict__ temp_all,
                                                          int *__restrict__ idx_all) {
    constexpr int T = NWAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // layout: uint2 slots[2][16] | float xs[n] ys[n] zs[n] (if XYZ_LDS)
    uint2 *slots = reinterpret_cast<uint2 *>(smem_raw);
    float *xs = reinterpret_cast<float *>(smem_raw + 2 * 16 * sizeof(uint2));
    float *ys = xs + n;
    float *zs = ys + n;

    const int b = blockIdx.x;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    float *temp = temp_all + (size_t)b * n;
    int *idx = idx_all + (size_t)b * m;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    float px[PPT], py[PPT], pz[PPT], dmin[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int k = tid + i * T;
        if (k < n) {
            px[i] = xyz[(size_t)k * 3 + 0];
            py[i] = xyz[(size_t)k * 3 + 1];
            pz[i] = xyz[(size_t)k * 3 + 2];
            dmin[i] = temp[k];
            if (XYZ_LDS) {
                xs[k] = px[i];
                ys[k] = py[i];
                zs[k] = pz[i];
            }
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            dmin[i] = 0.f;
        }
    }
    if (tid == 0) idx[0] = 0;
    if (XYZ_LDS) __syncthreads();

    int old = 0;
    for (int j = 1; j < m; ++j) {
        float ox, oy, oz;
        if (XYZ_LDS) {
            ox = xs[old];
            oy = ys[old];
            oz = zs[old];
        } else {
            ox = xyz[(size_t)old * 3 + 0];
            oy = xyz[(size_t)old * 3 + 1];
            oz = xyz[(size_t)old * 3 + 2];
        }
        unsigned bestd = 0u;          // bits of the best running-min distance of this lane
        unsigned besti = 0xFFFFFFFFu; // its index (lowest on ties: i ascends, strict '>')
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            int k = tid + i * T;
            float d = dist2_unfused(px[i], py[i], pz[i], ox, oy, oz);
            float d2 = fminf(d, dmin[i]);
            dmin[i] = d2;
            unsigned bits = __float_as_uint(d2);
            bool take = (k < n) && (besti == 0xFFFFFFFFu || bits > bestd);
            bestd = take ? bits : bestd;
            besti = take ? (unsigned)k : besti;
        }
        // wave stage
        unsigned wmax = wave_max_u32(bestd);
        unsigned cand = (bestd == wmax) ? besti : 0xFFFFFFFFu;
        unsigned wmin = wave_min_u32(cand);
        unsigned sel;
        if (NWAVES == 1) {
            sel = wmin;
        } else {
            uint2 *slot = slots + (j & 1) * 16;
            if (lane == 0) slot[wave] = make_uint2(wmax, wmin);
            __syncthreads();
            uint2 kv = slot[lane & (NWAVES - 1)];
            unsigned gmax = row_max_u32(kv.x);
            unsigned c2 = (kv.x == gmax) ? kv.y : 0xFFFFFFFFu;
            unsigned gmin = row_min_u32(c2);
            sel = (unsigned)__builtin_amdgcn_readfirstlane((int)gmin);
        }
        old = (int)sel;
        if (tid == 0) idx[j] = old;
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int k = tid + i * T;
        if (k < n) temp[k] = dmin[i];
    }
}

// Second generation of the register-resident kernel (the default): BLOCKED ownership -- lane l of wave w holds the PPT
// consecutive points starting at (64w + l) * PPT -- so that "lowest index among equal maxima" is "lowest wave, then
// lowest lane, then lowest slot": after one DPP max reduction the winner is picked with a ballot + find-first-set +
// v_readlane instead of a second 6-step reduction (and likewise across waves).  Distances are computed two points
// per instruction (v_pk_add_f32 / v_pk_mul_f32: same IEEE single operations, unfused), the running minimum and the
// argmax run on the non-negative floats' bit patterns (v_min_u32 / v_max3_u32: no NaN-canonicalising extras).
typedef fps_f32x2 f32x2;

template <int NWAVES, int PPT, bool XYZ_LDS>
__global__ __launch_bounds__(NWAVES * 64) void fps_kernel_blocked(int n_stride, int m, const float *__restrict__ xyz_all,
                                                                  float *__restrict__ temp_all, int *__restrict__ idx_all,
                                                                  float *__restrict__ new_n3, float *__restrict__ new_cn,
                                                                  const int *__restrict__ n_per_cloud, int defer, int j0, int j1) {
    // picks [j0, j1) of the m (captra_fps_gather_part): j0 > 0 continues a cloud's sampling -- the running minima come from `temp`
    // (what the previous part left there: distances to every pick before idx[j0 - 1]) and the round of pick j0 starts from idx[j0 - 1];
    // a part that starts at 0 with `temp` given initialises it itself.  The whole sampling is (0, m).
    static_assert(PPT % 2 == 0, "two points per packed instruction");
    constexpr int H = PPT / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint2 *slots = reinterpret_cast<uint2 *>(smem_raw);                  // [2][16] ping-pong
    float4 *slotc = reinterpret_cast<float4 *>(smem_raw + 2 * 16 * sizeof(uint2));  // !XYZ_LDS: the candidates' coordinates
    float *xs = reinterpret_cast<float *>(smem_raw + 2 * 16 * sizeof(uint2) + (XYZ_LDS ? 0 : 2 * 16 * sizeof(float4)));
    float *ys = xs + n_stride;
    float *zs = ys + n_stride;
    // defer (XYZ_LDS only, LDS permitting): the picks are parked in LDS and every output -- idx, the gathered coordinates in
    // both layouts -- is written after the last round by all threads; a round then issues no global store and no exec-masked
    // address arithmetic at all (seven single-lane stores per round before)
    int *picks = reinterpret_cast<int *>(zs + n_stride);
    const int b = blockIdx.x;
    // ragged batches: clouds padded to n_stride points, cloud b has n_per_cloud[b] of them (slots beyond n never win)
    const int n = n_per_cloud != nullptr ? n_per_cloud[b] : n_stride;
    const float *xyz = xyz_all + (size_t)b * n_stride * 3;
    float *temp = temp_all != nullptr ? temp_all + (size_t)b * n_stride : nullptr;  // null: start from 1e10, nothing written back
    int *idx = idx_all + (size_t)b * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = tid * PPT;
    const bool part = !(j0 == 0 && j1 == m);
    // optional gather of the sampled coordinates (pointnet_utils.py:222-223 index_points(xyz, fps_idx)): both layouts
    auto emit = [&](int j, float x, float y, float z) {
        if (new_n3 != nullptr) {
            float *d = new_n3 + ((size_t)b * m + j) * 3;
            d[0] = x; d[1] = y; d[2] = z;
        }
        if (new_cn != nullptr) {
            float *d = new_cn + (size_t)b * 3 * m + j;
            d[0] = x; d[m] = y; d[2 * (size_t)m] = z;
        }
    };

    f32x2 px[H], py[H], pz[H];
    unsigned dmin[PPT];   // running min distance, as bits (>= 0: bit order == value order)
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = base + i;
        float x = 0.f, y = 0.f, z = 0.f;
        unsigned d0 = 0u;                      // slots beyond n: distance 0 forever, highest indices -> never preferred
        if (k < n) {
            x = xyz[(size_t)k * 3 + 0]; y = xyz[(size_t)k * 3 + 1]; z = xyz[(size_t)k * 3 + 2];
            d0 = __float_as_uint(temp != nullptr && !(part && j0 == 0) ? temp[k] : 1e10f);
            if (XYZ_LDS) { xs[k] = x; ys[k] = y; zs[k] = z; }
        }
        px[i / 2][i & 1] = x; py[i / 2][i & 1] = y; pz[i / 2][i & 1] = z;
        dmin[i] = d0;
    }
    if (tid == 0 && j0 == 0) {
        if (XYZ_LDS && defer) picks[0] = 0;
        else idx[0] = 0;
    }
    __syncthreads();

    int old = j0 > 0 ? idx[j0 - 1] : 0;
    // clouds beyond the LDS budget: the winner's coordinates travel with its (distance, index) through the wave slots
    // (picked out of the owner lane's registers with a uniform slot switch + v_readlane), so a round reads no memory at all
    float nx = xyz[(size_t)old * 3 + 0], ny = xyz[(size_t)old * 3 + 1], nz = xyz[(size_t)old * 3 + 2];
    for (int j = j0 > 0 ? j0 : 1; j < j1; ++j) {
        // the last pick's coordinates: broadcast read of the LDS mirror, or the values carried over from the last round
        const float ox = XYZ_LDS ? xs[old] : nx;
        const float oy = XYZ_LDS ? ys[old] : ny;
        const float oz = XYZ_LDS ? zs[old] : nz;
        if (!(XYZ_LDS && defer) && tid == 0) emit(j - 1, ox, oy, oz);
        unsigned best = 0u;
        int li = PPT - 1;                       // lowest slot of this lane holding `best`
        if constexpr (PPT == 16) {
            fps_lane_round16(px, py, pz, dmin, ox, oy, oz, best, li);      // (fps_round.h: shared with the level-1 stream kernel)
        } else {
            fps_lane_round<PPT>(px, py, pz, dmin, ox, oy, oz, best, li);
        }
        const unsigned wmax = wave_max_u32_fold(best);
        const unsigned long long hit = __ballot(best == wmax);
        const int wl = __ffsll((long long)hit) - 1;            // lowest lane = lowest indices of the wave
        const unsigned widx = (unsigned)__builtin_amdgcn_readlane(base + li, wl);
        float wx = 0.f, wy = 0.f, wz = 0.f;
        if (!XYZ_LDS) {
            const int ls = __builtin_amdgcn_readlane(li, wl);  // the winner lane's slot: wave-uniform
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                if (ls == i) {                                  // uniform branch: exactly one case runs
                    wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(px[i / 2][i & 1]), wl));
                    wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(py[i / 2][i & 1]), wl));
                    wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(pz[i / 2][i & 1]), wl));
                }
        }
        unsigned sel;
        if (NWAVES == 1) {
            sel = widx;
            nx = wx; ny = wy; nz = wz;
        } else {
            uint2 *slot = slots + (j & 1) * 16;
            float4 *sc = slotc + (j & 1) * 16;
            if (lane == 0) {
                slot[wave] = make_uint2(wmax, widx);
                if (!XYZ_LDS) sc[wave] = make_float4(wx, wy, wz, 0.f);
            }
            __syncthreads();
            if constexpr (NWAVES == 4 && XYZ_LDS) {
                // four waves: every lane reads the four (maximum, index) pairs as two 16-byte broadcasts and picks the winner
                // with three strict compares in wave order (ties: the lower wave = the lower indices) -- no cross-lane
                // reduction, ballot or readlane on the round's critical chain; the pick stays in a (uniform) vector register
                old = fps_winner_of_four(slot);
                if (defer) {
                    if (wave == 0) picks[j - j0] = old;
                } else if (tid == 0) idx[j] = old;
                continue;
            }
            const uint2 kv = slot[lane & (NWAVES - 1)];
            float4 kc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!XYZ_LDS) kc = sc[lane & (NWAVES - 1)];
            const unsigned gmax = row_max_u32_fold(kv.x);
            const unsigned long long hit2 = __ballot(kv.x == gmax);
            const int gl = __ffsll((long long)hit2) - 1;       // lowest lane = lowest wave = lowest indices
            sel = (unsigned)__builtin_amdgcn_readlane((int)kv.y, gl);
            if (!XYZ_LDS) {
                nx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(kc.x), gl));
                ny = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(kc.y), gl));
                nz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(kc.z), gl));
            }
        }
        old = (int)sel;
        if (XYZ_LDS && defer) {
            if (wave == 0) picks[j - j0] = old;                 // (uniform branch; the 64 lanes store one word)
        } else if (tid == 0) idx[j] = old;
    }
    if (XYZ_LDS && defer) {
        __syncthreads();
        for (int j = j0 + tid; j < j1; j += NWAVES * 64) {
            const int id = picks[j - j0];
            idx[j] = id;
            emit(j, xs[id], ys[id], zs[id]);
        }
    } else if (tid == 0) emit(j1 - 1, xyz[(size_t)old * 3 + 0], xyz[(size_t)old * 3 + 1], xyz[(size_t)old * 3 + 2]);
    if (temp != nullptr) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int k = base + i;
            if (k < n) temp[k] = __uint_as_float(dmin[i]);
        }
    }
}

// Fallback for clouds too large for the register-resident kernel: running min in global memory,
// same selection rule.  1024 threads, strided ownership.
__global__ __launch_bounds__(1024) void fps_kernel_big(int n, int m, const float *__restrict__ xyz_all,
                                                       float *__restrict__ temp_all,
                                                       int *__restrict__ idx_all) {
    __shared__ uint2 slots[2][16];
    const int b = blockIdx.x;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    float *temp = temp_all + (size_t)b * n;
    int *idx = idx_all + (size_t)b * m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) idx[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        float ox = xyz[(size_t)old * 3 + 0], oy = xyz[(size_t)old * 3 + 1], oz = xyz[(size_t)old * 3 + 2];
        unsigned bestd = 0u, besti = 0xFFFFFFFFu;
        for (int k = tid; k < n; k += 1024) {
            float d = dist2_unfused(xyz[(size_t)k * 3 + 0], xyz[(size_t)k * 3 + 1], xyz[(size_t)k * 3 + 2], ox, oy, oz);
            float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            unsigned bits = __float_as_uint(d2);
            bool take = (besti == 0xFFFFFFFFu || bits > bestd);
            bestd = take ? bits : bestd;
            besti = take ? (unsigned)k : besti;
        }
        unsigned wmax = wave_max_u32(bestd);
        unsigned cand = (bestd == wmax) ? besti : 0xFFFFFFFFu;
        unsigned wmin = wave_min_u32(cand);
        if (lane == 0) slots[j & 1][wave] = make_uint2(wmax, wmin);
        __syncthreads();
        uint2 kv = slots[j & 1][lane & 15];
        unsigned gmax = row_max_u32(kv.x);
        unsigned c2 = (kv.x == gmax) ? kv.y : 0xFFFFFFFFu;
        unsigned gmin = row_min_u32(c2);
        old = (int)__builtin_amdgcn_readfirstlane((int)gmin);
        if (tid == 0) idx[j] = old;
    }
}

static CAPTRA_KNOB int g_fps_defer = 1;    // blocked kernel: outputs written after the last round (0 = inside the rounds, the first form)
static CAPTRA_KNOB int g_fps_variant = 0;  // 0 = blocked / ballot / packed-math kernel where it applies, 1 = first-generation kernel

template <int NWAVES, int PPT>
int launch_fps(int b, int n, int m, const float *xyz, float *temp, int *idx, hipStream_t s, float *new_n3 = nullptr,
               float *new_cn = nullptr, bool need_blocked = false, const int *ns = nullptr, int j0 = 0, int j1 = -1) {
    if (j1 < 0) j1 = m;
    size_t slots = 2 * 16 * sizeof(uint2);
    size_t lds_xyz = (size_t)n * 3 * sizeof(float);
    if constexpr (PPT % 2 == 0) {
        if (g_fps_variant == 0 && slots + lds_xyz <= 150 * 1024) {
            auto kern2 = fps_kernel_blocked<NWAVES, PPT, true>;
            static CaptraDeviceOnce once2;
            if (once2.first_use()) {
                hipFuncSetAttribute(reinterpret_cast<const void *>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                once2.done();
            }
            const size_t picks_b = (size_t)(j1 - j0) * sizeof(int);
            const int defer = (g_fps_defer && slots + lds_xyz + picks_b <= 150 * 1024) ? 1 : 0;
            CAPTRA_LAUNCH("fps", kern2, dim3(b), dim3(NWAVES * 64), slots + lds_xyz + (defer ? picks_b : 0), s, n, m, xyz, temp, idx, new_n3, new_cn, ns, defer, j0, j1);
            return captra_last_error();
        }
        if (g_fps_variant == 0) {  // cloud larger than the LDS mirror: same kernel, winner coordinates from global memory
            CAPTRA_LAUNCH("fps", (fps_kernel_blocked<NWAVES, PPT, false>), dim3(b), dim3(NWAVES * 64),
                          slots + 2 * 16 * sizeof(float4), s, n, m, xyz, temp, idx, new_n3, new_cn, ns, 0, j0, j1);
            return captra_last_error();
        }
    }
    if (need_blocked || ns != nullptr || j0 != 0 || j1 != m) return -2;  // the fused sample + gather entry (and its parts) exist on the blocked kernel only
    if (slots + lds_xyz <= 150 * 1024) {
        auto kern = fps_kernel<NWAVES, PPT, true>;
        static CaptraDeviceOnce once;
        if (once.first_use()) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            once.done();
        }
        CAPTRA_LAUNCH("fps", kern, dim3(b), dim3(NWAVES * 64), slots + lds_xyz, s, n, m, xyz, temp, idx);
    } else {
        CAPTRA_LAUNCH("fps", (fps_kernel<NWAVES, PPT, false>), dim3(b), dim3(NWAVES * 64), slots, s, n, m,
                      xyz, temp, idx);
    }
    return captra_last_error();
}

}  // namespace

// fps_pruned.hip: exact spatially-pruned kernel for 8k-20k point clouds (-2 beyond its capacity)
int captra_fps_pruned_launch(int b, int n_stride, const int *n_per_cloud, int m, const float *xyz, float *temp, int *idx,
                             float *new_n3, float *new_cn, hipStream_t s);
static CAPTRA_KNOB int g_fps_pruned_min = 8192;   // clouds of at least this many points take the pruned kernel (0 = never)
extern "C" void captra_fps_set_pruned_min(int n) { g_fps_pruned_min = n; }

// Tunable from the host for experiments: waves per cloud for the register-resident kernel
// (0 = heuristic).  Not part of the stable ABI.
static CAPTRA_KNOB int g_fps_waves = 0;
extern "C" void captra_fps_set_waves(int w) { g_fps_waves = w; }
extern "C" void captra_fps_set_variant(int v) { g_fps_variant = v; }
extern "C" void captra_fps_set_defer(int v) { g_fps_defer = v; }

extern "C" int captra_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                              int *idx, captra_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return -1;
    if (b == 0 || m == 0) return 0;
    if (n == 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (g_fps_variant == 0 && g_fps_waves == 0 && g_fps_pruned_min > 0 && n >= g_fps_pruned_min && m > 1) {
        const int rc = captra_fps_pruned_launch(b, n, nullptr, m, xyz, temp, idx, nullptr, nullptr, s);
        if (rc != -2) return rc;
    }
    int waves = g_fps_waves;
    if (waves == 0) {
        // heuristic: ~8 points per lane (blocked kernel) / ~4 (first generation), at most 16 waves
        const int per_lane = g_fps_variant == 0 ? 8 : 4;
        waves = 1;
        while (waves < 16 && waves * 64 * per_lane < n) waves *= 2;
    }
    int ppt = (n + waves * 64 - 1) / (waves * 64);
#define FPS_CASE(W, P) \
    if (waves == W && ppt <= P) return launch_fps<W, P>(b, n, m, xyz, temp, idx, s);
    FPS_CASE(1, 1) FPS_CASE(1, 2) FPS_CASE(1, 4) FPS_CASE(1, 8) FPS_CASE(1, 16)
    FPS_CASE(2, 1) FPS_CASE(2, 2) FPS_CASE(2, 4) FPS_CASE(2, 8) FPS_CASE(2, 16)
    FPS_CASE(4, 1) FPS_CASE(4, 2) FPS_CASE(4, 4) FPS_CASE(4, 8) FPS_CASE(4, 16)
    FPS_CASE(8, 1) FPS_CASE(8, 2) FPS_CASE(8, 4) FPS_CASE(8, 8) FPS_CASE(8, 16)
    FPS_CASE(16, 1) FPS_CASE(16, 2) FPS_CASE(16, 4) FPS_CASE(16, 8) FPS_CASE(16, 12) FPS_CASE(16, 16) FPS_CASE(16, 20)
    FPS_CASE(16, 24) FPS_CASE(16, 32)
#undef FPS_CASE
    CAPTRA_LAUNCH("fps", fps_kernel_big, dim3(b), dim3(1024), 0, s, n, m, xyz, temp, idx);
    return captra_last_error();
}

// FPS + gather of the sampled coordinates in one launch (see include/captra_hip.h).  -2 when the cloud does not fit the
// register-resident kernel (use captra_furthest_point_sampling + captra_gather_points then).
static int fps_gather_dispatch(int b, int n, const int *ns, int m, const float *xyz, int *idx, float *new_xyz_n3,
                               float *new_xyz_cn, hipStream_t s, float *temp = nullptr, int j0 = 0, int j1 = -1) {
    if (b < 0 || n < 1 || m < 1) return -1;
    if (b == 0) return 0;
    if (g_fps_variant != 0) return -2;
    if (j1 < 0) j1 = m;
    const bool part = !(j0 == 0 && j1 == m);
    if (part && (temp == nullptr || j0 < 0 || j1 > m || j0 >= j1 || ns != nullptr)) return -1;
    if (g_fps_pruned_min > 0 && n >= g_fps_pruned_min && m > 1) {
        if (part) return -2;                       // parts: the register-resident kernel only
        const int rc = captra_fps_pruned_launch(b, n, ns, m, xyz, nullptr, idx, new_xyz_n3, new_xyz_cn, s);
        if (rc != -2) return rc;
    }
    int waves = g_fps_waves;
    if (waves == 0) {
        waves = 1;
        while (waves < 16 && waves * 64 * 8 < n) waves *= 2;
        // 2049..4096 points: four waves (one per SIMD) x 16 points per lane instead of eight x 8 -- a round's fixed work
        // (wave reduction, exchange) is issued once per SIMD instead of twice: 285 -> 270 us for 4096 -> 512 at 16 and 32
        // clouds; two waves x 32 (385 us) and sixteen x 4 (328 us) lose
        if (waves == 8) waves = 4;
    }
    const int ppt = (n + waves * 64 - 1) / (waves * 64);
#define FPSG_CASE(W, P) \
    if (waves == W && ppt <= P) return launch_fps<W, P>(b, n, m, xyz, temp, idx, s, new_xyz_n3, new_xyz_cn, true, ns, j0, j1);
    FPSG_CASE(1, 2) FPSG_CASE(1, 4) FPSG_CASE(1, 8)
    FPSG_CASE(2, 8) FPSG_CASE(4, 8) FPSG_CASE(8, 8) FPSG_CASE(4, 16) FPSG_CASE(2, 32) FPSG_CASE(16, 4)   /* (the last three: captra_fps_set_waves experiments; 4 x 16 is the default for 2049..4096 points) */
    FPSG_CASE(16, 8) FPSG_CASE(16, 12) FPSG_CASE(16, 16) FPSG_CASE(16, 20)
    FPSG_CASE(16, 24) FPSG_CASE(16, 32)
#undef FPSG_CASE
    return -2;
}

extern "C" int captra_fps_gather(int b, int n, int m, const float *xyz, int *idx, float *new_xyz_n3, float *new_xyz_cn,
                                 captra_stream_t stream) {
    return fps_gather_dispatch(b, n, nullptr, m, xyz, idx, new_xyz_n3, new_xyz_cn, (hipStream_t)stream);
}

// Picks [j0, j1) of captra_fps_gather's m, as a launch of its own: state (B,N) fp32 carries a cloud's running minima from part
// to part (written by every part, read by every part but the one that starts at 0), idx / new_xyz_* are the whole sampling's
// buffers.  Parts launched in order on one stream produce what the one launch produces, bit for bit; what is between them is the
// caller's -- e.g. the ball query and shared MLPs of the centres picked so far, on other streams.  -2: cloud outside the
// register-resident kernel.
extern "C" int captra_fps_gather_part(int b, int n, int m, int j0, int j1, const float *xyz, float *state, int *idx, float *new_xyz_n3,
                                      float *new_xyz_cn, captra_stream_t stream) {
    if (j0 == 0 && j1 == m) return fps_gather_dispatch(b, n, nullptr, m, xyz, idx, new_xyz_n3, new_xyz_cn, (hipStream_t)stream);
    return fps_gather_dispatch(b, n, nullptr, m, xyz, idx, new_xyz_n3, new_xyz_cn, (hipStream_t)stream, state, j0, j1);
}

// Ragged batch: clouds padded to n_stride points each, cloud i samples from its first n_per_cloud[i] (device array).
extern "C" int captra_fps_gather_ragged(int b, int n_stride, const int *n_per_cloud, int m, const float *xyz, int *idx,
                                        float *new_xyz_n3, float *new_xyz_cn, captra_stream_t stream) {
    return fps_gather_dispatch(b, n_stride, n_per_cloud, m, xyz, idx, new_xyz_n3, new_xyz_cn, (hipStream_t)stream);
}
