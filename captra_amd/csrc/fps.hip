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

template <int NWAVES, int PPT>
int launch_fps(int b, int n, int m, const float *xyz, float *temp, int *idx, hipStream_t s) {
    size_t slots = 2 * 16 * sizeof(uint2);
    size_t lds_xyz = (size_t)n * 3 * sizeof(float);
    if (slots + lds_xyz <= 150 * 1024) {
        auto kern = fps_kernel<NWAVES, PPT, true>;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        CAPTRA_LAUNCH("fps", kern, dim3(b), dim3(NWAVES * 64), slots + lds_xyz, s, n, m, xyz, temp, idx);
    } else {
        CAPTRA_LAUNCH("fps", (fps_kernel<NWAVES, PPT, false>), dim3(b), dim3(NWAVES * 64), slots, s, n, m,
                      xyz, temp, idx);
    }
    return captra_last_error();
}

}  // namespace

// Tunable from the host for experiments: waves per cloud for the register-resident kernel
// (0 = heuristic).  Not part of the stable ABI.
static int g_fps_waves = 0;
extern "C" void captra_fps_set_waves(int w) { g_fps_waves = w; }

extern "C" int captra_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                              int *idx, captra_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return -1;
    if (b == 0 || m == 0) return 0;
    if (n == 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    int waves = g_fps_waves;
    if (waves == 0) {
        // heuristic: ~4 points per lane, at most 16 waves
        waves = 1;
        while (waves < 16 && waves * 64 * 4 < n) waves *= 2;
    }
    int ppt = (n + waves * 64 - 1) / (waves * 64);
#define FPS_CASE(W, P) \
    if (waves == W && ppt <= P) return launch_fps<W, P>(b, n, m, xyz, temp, idx, s);
    FPS_CASE(1, 1) FPS_CASE(1, 2) FPS_CASE(1, 4) FPS_CASE(1, 8) FPS_CASE(1, 16)
    FPS_CASE(2, 1) FPS_CASE(2, 2) FPS_CASE(2, 4) FPS_CASE(2, 8) FPS_CASE(2, 16)
    FPS_CASE(4, 1) FPS_CASE(4, 2) FPS_CASE(4, 4) FPS_CASE(4, 8) FPS_CASE(4, 16)
    FPS_CASE(8, 1) FPS_CASE(8, 2) FPS_CASE(8, 4) FPS_CASE(8, 8) FPS_CASE(8, 16)
    FPS_CASE(16, 1) FPS_CASE(16, 2) FPS_CASE(16, 4) FPS_CASE(16, 8) FPS_CASE(16, 16) FPS_CASE(16, 32)
#undef FPS_CASE
    CAPTRA_LAUNCH("fps", fps_kernel_big, dim3(b), dim3(1024), 0, s, n, m, xyz, temp, idx);
    return captra_last_error();
}
