// Furthest point sampling of LARGE clouds (8k - 20k points) with exact spatial pruning, for gfx950.
//
// The register-resident kernel of fps.hip updates every running minimum in every round: at 16384 -
// 20480 points (the on-the-fly re-crop of real NOCS tracking samples 4096 of up to 20480 candidates,
// reference data_utils.py:138-157; BASELINE.json configs[4] samples 2048 of 16384) that is ~1.5 us of
// pure VALU work per round on the one CU a cloud can use.  But a new sample s can only lower the
// running minimum of points closer to it than their current minimum, i.e. (after the first few
// rounds) of a small neighbourhood.  This kernel
//   1. sorts the cloud by a 12-bit Morton cell (one LDS counting sort, in the workgroup),
//   2. cuts the sorted sequence into BUCKETS of 64 consecutive points — one point per lane, bucket k
//      owned by wave k % NW (interleaved, so that the buckets a sample touches spread over the waves)
//      — and keeps per bucket its bounding box, the largest running minimum `bmax`, the sorted
//      position of the point holding it and that point's coordinates (lane i of a few VGPRs = bucket i
//      of the wave),
//   3. per round tests all buckets of a wave at once: lb = squared distance from s to the bucket's box,
//      computed with the SAME unfused fp32 operations and association as the point distances, so by
//      monotonicity of rounding lb <= d(s, p) for every point p of the bucket; lb >= bmax  =>  no
//      running minimum of the bucket changes and the bucket is skipped.  The buckets that fail the test
//      are updated one by one (uniform branch into the slot's registers).
// The selection is then a reduction over bucket maxima.  Picks are IDENTICAL to the plain kernel's:
// same arithmetic for every distance that can matter, and "lowest original index among equal maxima"
// is resolved exactly (the original index of a sorted position is kept in LDS and consulted whenever
// a maximum is attained more than once — always the case in duplicate-padded clouds).
//
// SEVERAL PICKS PER ROUND (round 3).  A round costs ~420 instructions per wave whatever it decides (bucket tests, wave
// candidate, barrier, exchange) and the CU is issue-bound on them (two waves per SIMD in lock step), and there were m - 1
// rounds.  But the NEXT pick is often known before the current one is applied: let a be this round's pick and c the best
// remaining candidate.  Every running minimum only ever decreases, so after the update with a no point can exceed its old
// value; if c itself is untouched by a -- dist(c, a) >= dmin(c) in the very fp32 arithmetic the update would use -- and
// positive, c is still the maximum (ties: c was the lowest original index among equals before, and whoever equals it
// afterwards equalled it before), i.e. c IS the next pick, and so on down the sorted list.  A wave knows only its TOP point,
// so the list is the NW wave tops, and a wave whose top was taken contributes an OBSTACLE instead: an upper bound on
// everything else it holds (its second-largest bucket maximum, and the runner-up of the top's own bucket, kept per bucket
// next to its maximum).  A candidate must beat, strictly, the obstacles of the waves already picked from.  The certification
// runs on the 8 x 8 pairs of wave tops, one pair per lane, after the one barrier of the round (two ballots give every top
// its predecessors and whether one of them spoils it); the picks are then applied together, each touched bucket read and
// written once for all of them.  Picks are identical to the one-per-round kernel's (the same tests pass, incl.
// duplicate-padded and all-identical clouds).  Up to KP = 4 per round: 4095 picks take ~1440 rounds on the re-crop's clouds
// (2.8 per round); KP = 8 certifies 3.1 per round but its longer sample loops cost more than they save.  Measured against
// the one-per-round kernel in the same process (tools/bench_fps.py): 15000 -> 4096 2.86 -> 2.58 ms, 20480 -> 4096
// 3.47 -> 3.09 ms, 16384 -> 2048 1.53 -> 1.46 ms on surface-like clouds; uniform clouds -2 .. -7 %, except 16384 uniform
// points (+6 %: every slot of every wave in use, and a round now waits for the wave with the most buckets to update).
#include "common.h"
#include <type_traits>

namespace {

constexpr int FP_BINS = 4096;       // 16 x 16 x 16 Morton cells

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned fp_dpp_max(unsigned v) {   // 0 is the identity: invalid sources contribute 0
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    return max(v, o);
}
__device__ __forceinline__ unsigned fp_row_max(unsigned v) {
    v = fp_dpp_max<0xB1, 0xF>(v);
    v = fp_dpp_max<0x4E, 0xF>(v);
    v = fp_dpp_max<0x141, 0xF>(v);
    v = fp_dpp_max<0x140, 0xF>(v);
    return v;
}
__device__ __forceinline__ unsigned fp_wave_max(unsigned v) {
    v = fp_row_max(v);
    v = fp_dpp_max<0x142, 0xA>(v);
    v = fp_dpp_max<0x143, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned fp_wave_min(unsigned v) { return ~fp_wave_max(~v); }
__device__ __forceinline__ float fp_wave_minf(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float fp_wave_maxf(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float fp_readlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ unsigned fp_spread4(unsigned v) {   // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

// A floats per lane held in REGISTERS yet indexed at run time by a wave-uniform slot number: an ext-vector value, which
// the backend addresses through the GPR index mode (s_set_gpr_idx_on + v_mov) instead of spilling an alloca.  (Legal
// widths are 8 / 16 / 32; 8-wide vectors are expanded into a compare / select chain instead — measured 0.3 us slower
// per round — so slots beyond 32 live in LDS, see below.)
template <int A>
struct RegVec {
    typedef float VA __attribute__((ext_vector_type(A)));
    VA a;
    __device__ __forceinline__ float get(int s) const { return a[s]; }
    __device__ __forceinline__ void set(int s, float v) { a[s] = v; }
};

// SLOTS = A + B buckets per wave: the first A in registers, B more in LDS (x, y, z and the running minimum of a bucket
// as four rows of 64 floats); capacity = FP_NW * SLOTS * 64 points.
template <int FP_NW, int A, int B, bool STATS>
__global__ __launch_bounds__(FP_NW * 64) void fps_pruned_kernel(int n_stride, const int *__restrict__ n_per_cloud, int m,
                                                          const float *__restrict__ xyz_all, float *__restrict__ temp_all,
                                                          int *__restrict__ idx_all, float *__restrict__ new_n3,
                                                          float *__restrict__ new_cn, unsigned long long *stats) {
    constexpr int SLOTS = A + B;
    constexpr int FP_T = FP_NW * 64, FP_BPT = FP_BINS / FP_T;
    constexpr int CAP = FP_NW * SLOTS * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(SLOTS <= 64, "one metadata lane per bucket");
    uint2 *slots = reinterpret_cast<uint2 *>(smem_raw);                                   // [2][16] (bmax, sorted pos)
    float4 *slotc = reinterpret_cast<float4 *>(smem_raw + 2 * 16 * sizeof(uint2));       // [2][16] the candidates' coordinates
    float *red = reinterpret_cast<float *>(smem_raw + 2 * 16 * (sizeof(uint2) + sizeof(float4)));   // [6][16] bbox partials
    float4 *picks = reinterpret_cast<float4 *>(red + 6 * 16);                             // [NW][8] a round's picks, one copy per wave
    unsigned short *oidx = reinterpret_cast<unsigned short *>(picks + FP_NW * 8);             // [CAP] original index of a sorted position (< 65535)
    unsigned char *scratch = reinterpret_cast<unsigned char *>(oidx + CAP);
    unsigned *hist = reinterpret_cast<unsigned *>(scratch);                               // [FP_BINS] (sort phase)
    unsigned *wsum = hist + FP_BINS;                                                      // [16]
    float *ovf = reinterpret_cast<float *>(scratch);                                      // [NW][B][4][64] (after the sort; reuses hist)

    const int b = blockIdx.x;
    const int n = n_per_cloud != nullptr ? n_per_cloud[b] : n_stride;
    if (n < 1 || n > n_stride) return;   // workgroup-uniform: nothing to sample from
    const float *xyz = xyz_all + (size_t)b * n_stride * 3;
    float *temp = temp_all != nullptr ? temp_all + (size_t)b * n_stride : nullptr;
    int *idx = idx_all + (size_t)b * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // ---- 1. bounding box of the cloud
    float lx = __builtin_inff(), ly = lx, lz = lx, hx = -lx, hy = -lx, hz = -lx;
    for (int k = tid; k < n; k += FP_T) {
        const float x = xyz[(size_t)k * 3 + 0], y = xyz[(size_t)k * 3 + 1], z = xyz[(size_t)k * 3 + 2];
        lx = fminf(lx, x); ly = fminf(ly, y); lz = fminf(lz, z);
        hx = fmaxf(hx, x); hy = fmaxf(hy, y); hz = fmaxf(hz, z);
    }
    lx = fp_wave_minf(lx); ly = fp_wave_minf(ly); lz = fp_wave_minf(lz);
    hx = fp_wave_maxf(hx); hy = fp_wave_maxf(hy); hz = fp_wave_maxf(hz);
    if (lane == 0) {
        red[0 * 16 + wave] = lx; red[1 * 16 + wave] = ly; red[2 * 16 + wave] = lz;
        red[3 * 16 + wave] = hx; red[4 * 16 + wave] = hy; red[5 * 16 + wave] = hz;
    }
    for (int e = tid; e < FP_BINS; e += FP_T) hist[e] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < FP_NW; ++w) {
        lx = fminf(lx, red[0 * 16 + w]); ly = fminf(ly, red[1 * 16 + w]); lz = fminf(lz, red[2 * 16 + w]);
        hx = fmaxf(hx, red[3 * 16 + w]); hy = fmaxf(hy, red[4 * 16 + w]); hz = fmaxf(hz, red[5 * 16 + w]);
    }
    const float sx = 16.f / fmaxf(hx - lx, 1e-30f), sy = 16.f / fmaxf(hy - ly, 1e-30f), sz = 16.f / fmaxf(hz - lz, 1e-30f);

    // ---- 2. counting sort by Morton cell: dst[i] = sorted position of this thread's i-th point
    unsigned dst[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int k = tid + i * FP_T;
        dst[i] = 0xFFFFFFFFu;
        if (k < n) {
            const float x = xyz[(size_t)k * 3 + 0], y = xyz[(size_t)k * 3 + 1], z = xyz[(size_t)k * 3 + 2];
            const int cx = min(15, max(0, (int)((x - lx) * sx))), cy = min(15, max(0, (int)((y - ly) * sy))),
                      cz = min(15, max(0, (int)((z - lz) * sz)));
            const unsigned key = fp_spread4((unsigned)cx) | (fp_spread4((unsigned)cy) << 1) | (fp_spread4((unsigned)cz) << 2);
            const unsigned rank = atomicAdd(&hist[key], 1u);
            dst[i] = (key << 20) | rank;           // rank < 2^20
        }
    }
    __syncthreads();
    {   // exclusive scan of the 4096 bins: FP_BPT bins per thread, wave scan, wave totals through LDS
        unsigned c[FP_BPT], tsum = 0u;
#pragma unroll
        for (int e = 0; e < FP_BPT; ++e) { c[e] = hist[tid * FP_BPT + e]; tsum += c[e]; }
        unsigned incl = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned run = incl - tsum;
#pragma unroll
        for (int w = 0; w < FP_NW; ++w) run += (w < wave) ? wsum[w] : 0u;
#pragma unroll
        for (int e = 0; e < FP_BPT; ++e) { hist[tid * FP_BPT + e] = run; run += c[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int k = tid + i * FP_T;
        if (k < n) {
            dst[i] = hist[dst[i] >> 20] + (dst[i] & 0xFFFFFu);
            oidx[dst[i]] = (unsigned short)k;
        }
    }
    for (int q = n + tid; q < CAP; q += FP_T) oidx[q] = 0xFFFFu;   // sentinels: never preferred
    __syncthreads();   // hist is dead from here on: its memory holds the LDS-resident slots

    // ---- 3. ownership: sorted position q -> wave (q / 64) % NW, slot (q / 64) / NW, lane q % 64.  Each lane fetches
    // its points through the permutation (a one-time 12-byte gather per point, served by L2).
    RegVec<A> px, py, pz, dmin;   // dmin: running minimum as a float (>= 0: value order == bit order)
    float *ovf_w = ovf + (size_t)wave * B * 256 + lane;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int q = ((s * FP_NW + wave) << 6) + lane;
        // slots past the end of the cloud: a copy of the last sorted point (so bounding boxes need no masking) whose
        // distance is 0 forever and whose original index is the sentinel — never selected
        const unsigned k = oidx[q < n ? q : n - 1];
        const float x = xyz[(size_t)k * 3 + 0], y = xyz[(size_t)k * 3 + 1], z = xyz[(size_t)k * 3 + 2];
        const float d0 = q < n ? (temp != nullptr ? temp[k] : 1e10f) : 0.f;
        if (s < A) {
            px.set(s, x); py.set(s, y); pz.set(s, z); dmin.set(s, d0);
        } else {
            float *o = ovf_w + (s - A) * 256;
            o[0] = x; o[64] = y; o[128] = z; o[192] = d0;
        }
    }

    // ---- 4. bucket metadata: lane i of these registers describes bucket (slot) i of this wave
    constexpr int KP = 4;   // picks per round, at most
    float blx = __builtin_inff(), bly = blx, blz = blx, bhx = -blx, bhy = -blx, bhz = -blx;   // lanes >= SLOTS: empty box
    unsigned bmax = 0u;
    unsigned brun = 0u;      // upper bound on the bucket's SECOND largest running minimum (refreshed with bmax; values only decrease)
    unsigned boi = 0xFFFFu;  // original index of the point holding bmax
    float bwx = 0.f, bwy = 0.f, bwz = 0.f;
    // the samples still to be applied (wave-uniform; position order of the previous round's picks)
    float ox[KP], oy[KP], oz[KP];
    int np = 1;
    // Update bucket s (wave-uniform, run time) against the np pending samples: the lane's point of that bucket is read
    // out of the register vectors by index, so ONE copy of this code serves every slot (an unrolled per-slot version
    // is ~90 KB of instructions and lost more to instruction fetch than the pruning saved).  A sample whose box test
    // excluded this bucket changes nothing here (lb <= d for every point of the bucket), so all of them are applied.
    unsigned n_upd = 0, n_ref = 0;   // experiment counters (captra_fps_set_stats)
    auto update = [&](int s, bool first) __attribute__((always_inline)) {
        ++n_upd;
        float x, y, z, dm;
        float *o = ovf_w + (s - A) * 256;
        if (B == 0 || s < A) { x = px.get(s); y = py.get(s); z = pz.get(s); dm = dmin.get(s); }
        else { x = o[0]; y = o[64]; z = o[128]; dm = o[192]; }
        const unsigned dold = __float_as_uint(dm);
        unsigned dnew = dold;
#pragma unroll
        for (int t = 0; t < KP; ++t) {
            if (t < np) {
                const float dx = x - ox[t], dy = y - oy[t], dz = z - oz[t];
                const unsigned d = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                dnew = d < dnew ? d : dnew;
            }
        }
        if (B == 0 || s < A) dmin.set(s, __uint_as_float(dnew));
        else o[192] = __uint_as_float(dnew);
        if (first) {   // the bucket's bounding box (once)
            const float a0 = fp_wave_minf(x), a1 = fp_wave_minf(y), a2 = fp_wave_minf(z);
            const float b0 = fp_wave_maxf(x), b1 = fp_wave_maxf(y), b2 = fp_wave_maxf(z);
            if (lane == s) { blx = a0; bly = a1; blz = a2; bhx = b0; bhy = b1; bhz = b2; }
        } else {
            // the bucket's maximum (and who holds it) can only change if a point AT the maximum was lowered
            const unsigned bm_old = (unsigned)__builtin_amdgcn_readlane((int)bmax, s);
            if (__ballot(dnew != dold && dold == bm_old) == 0ull) return;
        }
        ++n_ref;
        const unsigned bm = fp_wave_max(dnew);
        const unsigned long long hit = __ballot(dnew == bm);
        int wl = __ffsll((long long)hit) - 1;
        const int q0 = (s * FP_NW + wave) << 6;
        const unsigned oi = (unsigned)oidx[q0 + lane];
        if (hit & (hit - 1)) {   // the maximum is attained more than once: lowest ORIGINAL index wins
            const unsigned best = fp_wave_min((dnew == bm) ? oi : 0xFFFFFFFFu);
            wl = __ffsll((long long)__ballot(dnew == bm && oi == best)) - 1;
        }
        const float wx = fp_readlane(x, wl), wy = fp_readlane(y, wl), wz = fp_readlane(z, wl);
        const unsigned woi = (unsigned)__builtin_amdgcn_readlane((int)oi, wl);
        const unsigned ru = fp_wave_max(lane == wl ? 0u : dnew);      // the rest of the bucket never exceeds this
        if (lane == s) { bmax = bm; boi = woi; bwx = wx; bwy = wy; bwz = wz; brun = ru; }
    };
    // test every bucket of the wave against the pending samples (same operations as the point distance: fl(p - o), squares,
    // (x + y) + z) and update those that can change
    auto apply_pending = [&]() __attribute__((always_inline)) {
        unsigned long long need = 0ull;
#pragma unroll
        for (int t = 0; t < KP; ++t) {
            if (t < np) {
                const float ddx = fmaxf(fmaxf(blx - ox[t], ox[t] - bhx), 0.f), ddy = fmaxf(fmaxf(bly - oy[t], oy[t] - bhy), 0.f),
                            ddz = fmaxf(fmaxf(blz - oz[t], oz[t] - bhz), 0.f);
                const float lb = (ddx * ddx + ddy * ddy) + ddz * ddz;
                need |= __ballot(lane < SLOTS && __float_as_uint(lb) < bmax);
            }
        }
        while (need) {
            const int s = __ffsll((long long)need) - 1;
            need &= need - 1;
            update(s, false);
        }
    };
    // initial maxima: an "update" against a sample at infinity leaves every running minimum as it is
#pragma unroll
    for (int t = 0; t < KP; ++t) { ox[t] = __builtin_inff(); oy[t] = ox[t]; oz[t] = ox[t]; }
    const int slots_used = (n + FP_T - 1) / FP_T;   // buckets beyond the cloud keep bmax = 0 and an empty box: never touched
    for (int s = 0; s < slots_used; ++s) update(s, true);

    // ---- 5. the selection rounds: up to KP certified picks each (header)
    static_assert(FP_NW * FP_NW <= 64 && (FP_NW & (FP_NW - 1)) == 0, "pairs of wave tops on the lanes of one wave");
    if (tid == 0) { idx[0] = 0; emit(0, xyz[0], xyz[1], xyz[2]); }
    ox[0] = xyz[0]; oy[0] = xyz[1]; oz[0] = xyz[2];
    unsigned long long tph[4] = {0ull, 0ull, 0ull, 0ull};   // phase cycles (stats only)
    int rounds = 0;
    for (int j = 1; j < m;) {
        const unsigned long long t0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        apply_pending();
        const unsigned long long t1 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        // the wave's candidate (its largest bucket maximum, lowest original index among equals) and its obstacle: nothing
        // else this wave holds exceeds max(second-largest bucket maximum, runner-up bound of the candidate's bucket)
        const unsigned wm = fp_wave_max(bmax);
        const unsigned long long hitb = __ballot(bmax == wm);
        int li = __ffsll((long long)hitb) - 1;
        if (hitb & (hitb - 1)) {
            const unsigned best = fp_wave_min((bmax == wm) ? boi : 0xFFFFFFFFu);
            li = __ffsll((long long)__ballot(bmax == wm && boi == best)) - 1;
        }
        const unsigned second = fp_wave_max(lane == li ? 0u : bmax);
        const unsigned topru = (unsigned)__builtin_amdgcn_readlane((int)brun, li);
        const unsigned obst = second > topru ? second : topru;
        uint2 *slot = slots + (rounds & 1) * 16;
        float4 *sc = slotc + (rounds & 1) * 16;
        if (lane == li) {
            slot[wave] = make_uint2(bmax, boi);
            sc[wave] = make_float4(bwx, bwy, bwz, __uint_as_float(obst));
        }
        const unsigned long long t2 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        __syncthreads();
        const unsigned long long t3 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        // every wave certifies the picks of this round, lane (i, w) = (lane / NW, lane % NW) comparing wave top i with wave
        // top w: in the order (value descending, original index ascending) a top is picked iff everything before it is
        // picked, it is positive, no sample before it would lower it, and it beats (strictly) what those samples' waves
        // still hold.  (NW * NW <= 64.)
        const int ci = (lane / FP_NW) & (FP_NW - 1), cw = lane & (FP_NW - 1);
        const uint2 av = slot[ci], kv = slot[cw];
        const float4 ac = sc[ci], kc = sc[cw];
        const bool bf = kv.x > av.x || (kv.x == av.x && (kv.y < av.y || (kv.y == av.y && cw < ci)));
        const float pdx = ac.x - kc.x, pdy = ac.y - kc.y, pdz = ac.z - kc.z;
        const unsigned pd = __float_as_uint((pdx * pdx + pdy * pdy) + pdz * pdz);
        const bool in_pairs = lane < FP_NW * FP_NW;
        const unsigned long long bm64 = __ballot(in_pairs && bf);
        const unsigned long long bad64 = __ballot(in_pairs && bf && (pd < av.x || __float_as_uint(kc.w) >= av.x));
        // from here on lane l speaks for wave top l % NW
        const unsigned before = (unsigned)(bm64 >> (FP_NW * cw)) & ((1u << FP_NW) - 1u);
        const unsigned bad = (unsigned)(bad64 >> (FP_NW * cw)) & ((1u << FP_NW) - 1u);
        const int pos = __popc(before);
        const bool ok = pos == 0 || (bad == 0u && kv.x > 0u);
        const unsigned okmask = (unsigned)__ballot(ok && lane < FP_NW);
        const bool picked = lane < FP_NW && ok && (before & ~okmask) == 0u && pos < KP && j + pos < m;
        np = __popcll(__ballot(picked));
        float4 *mine = picks + wave * KP;        // wave-private: no barrier
        if (picked) {
            mine[pos] = kc;
            if (wave == 0) { idx[j + pos] = (int)kv.y; emit(j + pos, kc.x, kc.y, kc.z); }
        }
#pragma unroll
        for (int t = 0; t < KP; ++t) {
            const float4 pk = mine[t];           // (stale beyond np: never used)
            ox[t] = pk.x; oy[t] = pk.y; oz[t] = pk.z;
        }
        j += np;
        ++rounds;
        if (STATS) {
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2; tph[3] += t4 - t3;
        }
    }
    if (temp != nullptr && np > 1) {   // the running minima handed back have seen every sample but the last one
        np -= 1;
        apply_pending();
    }
    if (STATS && stats != nullptr && lane == 0) {   // per wave: bucket updates, maximum refreshes (both include the SLOTS initial ones)
        atomicAdd(stats + 0, (unsigned long long)n_upd);
        atomicAdd(stats + 1, (unsigned long long)n_ref);
        for (int i = 0; i < 4; ++i) atomicAdd(stats + 2 + i, tph[i]);   // test + updates | candidate | barrier wait | exchange
        atomicAdd(stats + 6, (unsigned long long)rounds);
    }
    if (temp != nullptr) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int q = ((s * FP_NW + wave) << 6) + lane;
            if (q < n) temp[oidx[q]] = s < A ? dmin.get(s) : ovf_w[(s - A) * 256 + 192];
        }
    }
}

unsigned long long *g_fps_stats = nullptr;

template <int FP_NW, int A, int B>
int launch_pruned(int b, int n_stride, const int *ns, int m, const float *xyz, float *temp, int *idx, float *new_n3,
                  float *new_cn, hipStream_t s) {
    constexpr int CAP = FP_NW * (A + B) * 64;
    const size_t sort_bytes = (size_t)(FP_BINS + 16) * sizeof(unsigned), ovf_bytes = (size_t)FP_NW * B * 256 * sizeof(float);
    const size_t shmem = 2 * 16 * (sizeof(uint2) + sizeof(float4)) + 6 * 16 * sizeof(float) + FP_NW * 8 * sizeof(float4) + (size_t)CAP * 2 +
                         (sort_bytes > ovf_bytes ? sort_bytes : ovf_bytes);
    constexpr int FP_T = FP_NW * 64;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(fps_pruned_kernel<FP_NW, A, B, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipFuncSetAttribute(reinterpret_cast<const void *>(fps_pruned_kernel<FP_NW, A, B, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        once.done();
    }
    if (g_fps_stats != nullptr) {   // instrumented build of the same kernel (counters + s_memtime per phase)
        CAPTRA_LAUNCH("fps", (fps_pruned_kernel<FP_NW, A, B, true>), dim3(b), dim3(FP_T), shmem, s, n_stride, ns, m, xyz, temp,
                      idx, new_n3, new_cn, g_fps_stats);
    } else {
        CAPTRA_LAUNCH("fps", (fps_pruned_kernel<FP_NW, A, B, false>), dim3(b), dim3(FP_T), shmem, s, n_stride, ns, m, xyz, temp,
                      idx, new_n3, new_cn, g_fps_stats);
    }
    return captra_last_error();
}

}  // namespace

// experiment hook: device pointer to six u64 counters accumulated by every wave: bucket updates, maximum refreshes,
// and s_memtime cycles of the four phases of a round (test + updates, wave candidate, barrier wait, exchange)
extern "C" void captra_fps_set_stats(unsigned long long *dev_counters) { g_fps_stats = dev_counters; }

// Internal entry used by fps.hip's dispatchers: -2 when the cloud exceeds the kernel's capacity (20480 points).
int captra_fps_pruned_launch(int b, int n_stride, const int *n_per_cloud, int m, const float *xyz, float *temp, int *idx,
                             float *new_n3, float *new_cn, hipStream_t s) {
    // 8 waves (2 per SIMD, 256 VGPRs each); 16 waves x 16 slots measured no faster (exchange and barrier cost more)
    // 32 register slots whatever the cloud size up to 16384 points (unused slots are never touched), 8 more in LDS beyond
    if (n_stride <= 8 * 64 * 32) return launch_pruned<8, 32, 0>(b, n_stride, n_per_cloud, m, xyz, temp, idx, new_n3, new_cn, s);
    if (n_stride <= 8 * 64 * 40) return launch_pruned<8, 32, 8>(b, n_stride, n_per_cloud, m, xyz, temp, idx, new_n3, new_cn, s);
    return -2;
}
