// f32x6 dense layer CHAINS for gfx950 (arithmetic contract: include/captra_hip.h "f32x6", csrc/sa_x6.hip): the 4096-point tails of
// the backbone in the opt-in arithmetic --
//   captra_mlp_chain3_x6:  y = act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3), the FP1 shared MLP + the backbone's conv1
//                          (reference network/models/pointnet_utils.py:296-298, backbones.py:66-68; exact form: csrc/mlp_chain.hip);
//   captra_coord_tail_x6:  the same three layers (feat, never stored) + CoordinateNet's segmentation head (one conv) + NOCS head
//                          (conv + BN + ReLU, conv, sigmoid - 0.5): six layers, two stored outputs (networks.py:29-32, 44-46,
//                          blocks.py:118-135).
// Same machine as the SA2 scales of csrc/sa_x6.hip: a wave (one per SIMD) carries 32 positions through every layer in registers
// (zero-swap hand-over: the accumulator tile is the next layer's operand under the weights' permuted k order; ReLU + three-way split
// as 12 VALU per register pair, pinned behind the next tile's MFMAs), the weights -- 304 / 448 KB of split fragment triples -- go
// round an LDS ring of three slots filled by LDS-DMA (one row tile per chunk, a quarter of the pieces per wave, one s_barrier per
// chunk), two accumulator chains per tile (hi / lo products).  What differs: the first layer is an x6 layer too (131 / 134 input
// channels: nine k-steps; its operand is the position's input column, loaded and split at the head of a slice), and the last
// layer STORES (128-byte row segments per half-wave) instead of reducing.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int cx_i32x4 __attribute__((ext_vector_type(4)));

constexpr int cx_cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cx_pad4(int a) { return (a + 3) / 4 * 4; }

__device__ __forceinline__ unsigned cx_pack(float lo, float hi) {          // one v_cvt_pk_bf16_f32 (RNE)
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float cx_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float cx_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void cx_split2(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
    p0 = cx_pack(a, b);
    const float ra = a - cx_lo(p0), rb = b - cx_hi(p0);
    p1 = cx_pack(ra, rb);
    p2 = cx_pack(ra - cx_lo(p1), rb - cx_hi(p1));
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));        // pinned where it is written (sa_x6.hip: LLVM sinks the split to its first use)
}
__device__ __forceinline__ f32x16 cx_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of one k-step as two accumulator chains issued alternately (sa_x6.hip: sx_group)
__device__ __forceinline__ void cx_group(f32x16 &hi, f32x16 &lo, const u32x4 (&w)[3], const u32x4 &x0, const u32x4 &x1, const u32x4 &x2) {
    lo = cx_mfma(w[2], x0, lo); hi = cx_mfma(w[1], x0, hi); lo = cx_mfma(w[0], x2, lo);
    hi = cx_mfma(w[0], x1, hi); lo = cx_mfma(w[1], x1, lo); hi = cx_mfma(w[0], x0, hi);
}
template <int I, int N, typename F>
__device__ __forceinline__ void cx_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cx_static_for<I + 1, N>(f);
    }
}
__host__ __device__ constexpr int cx_perm(int s) { return (s & 3) | ((s & 4) << 1) | ((s & 8) >> 1); }
constexpr int cx_unit_group(int u, int kst) { return kst > 1 ? 1 + u * (kst - 1) / 8 : 0; }

#define CX_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0F70)

// The layer table.  TAIL = false: three layers C0 -> 128 -> 128 -> 128 (the last stored).  TAIL = true: six -- C0 -> 128 -> 128 -> 128
// (feat), feat -> SEG (stored), feat -> 128 -> NOCS (stored).  Activations live in two register buffers: A (nine k-steps: the input
// column, then h2, then the NOCS head's hidden layer) and B (eight: h1, then feat).
template <int C0, bool TAIL>
struct CxShape {
    static constexpr int C = 128;
    static constexpr int NL = TAIL ? 6 : 3;
    static constexpr int KST0 = cx_cdiv(C0, 16);
    static constexpr int kst(int l) { return l == 0 ? KST0 : 8; }
    static constexpr int nt(int l) { return (TAIL && (l == 3 || l == 5)) ? 1 : 4; }
    static constexpr int frags(int l) { return cx_pad4(kst(l) * 3); }          // one chunk = one row tile of a layer
    static constexpr bool in_a(int l) { return l == 0 || l == 2 || l == 5; }    // the layer's operand buffer
    static constexpr bool out_a(int l) { return l == 1 || l == 4; }             // where a hidden layer's output goes
    static constexpr bool stores(int l) { return TAIL ? (l == 3 || l == 5) : l == 2; }
    static constexpr int tile0(int l) { int s = 0; for (int i = 0; i < l; ++i) s += nt(i); return s; }
    static constexpr int NCH = tile0(NL);
    static constexpr int layer_of(int tau) { int l = 0; while (tau >= tile0(l + 1)) ++l; return l; }
    static constexpr int chunk_off(int tau) {
        int off = 0;
        for (int i = 0; i < tau; ++i) off += frags(layer_of(i));
        return off * 1024;
    }
    static constexpr int WBYTES = chunk_off(NCH);
    static constexpr int SLOTB = frags(0) * 1024;
    static constexpr int group0(int tau) { int s = 0; for (int i = 0; i < tau; ++i) s += kst(layer_of(i)); return s; }
    static constexpr int G = group0(NCH);
    static constexpr int tile_of_group(int g) { int tau = 0; while (g >= group0(tau + 1)) ++tau; return tau; }
    static constexpr int NBIAS = NL * 128;                                      // floats behind the fragments: layer l at 128 l
};

// ---- image builder: one layer per launch --------------------------------------------------------------------------------------
// wt: the layer's packed fp32 W'^T (row-major part, element [k * ldw + cout]).  Fragment (tile t, k-step kk, part): lane l = row
// 32 t + (l & 31), k-slots 8 (l >> 5) .. + 7; slot s holds channel 16 kk + s (natural: the first layer, whose operand is loaded in
// that order) or 16 kk + perm(s) (the in-register hand-over order).
__global__ void pack_chain_x6_kernel(int cin, int cout, int ldw, int natural, const float *__restrict__ wt, const float *__restrict__ bias,
                                     unsigned char *__restrict__ img, float *__restrict__ bias_out) {
    const int kst = (cin + 15) / 16, nt = (cout + 31) / 32, ch = (kst * 3 + 3) / 4 * 4;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long long)nt * ch * 512) {
        const int f = (int)(e >> 9), lane = (int)(e >> 3) & 63, el = (int)e & 7;
        const int slot = 8 * (lane >> 5) + el;
        const int t = f / ch, r = f % ch, kk = r / 3, part = r % 3;
        const int row = 32 * t + (lane & 31), k = 16 * kk + (natural ? slot : cx_perm(slot));
        float v = 0.f;
        if (kk < kst && row < cout && k < cin) v = wt[(size_t)k * ldw + row];
        const __bf16 h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const __bf16 h2 = (__bf16)(r1 - (float)h1);
        const __bf16 hh = part == 0 ? h0 : (part == 1 ? h1 : h2);
        reinterpret_cast<unsigned short *>(img)[e] = __builtin_bit_cast(unsigned short, hh);
    } else {
        const int i = (int)(e - (long long)nt * ch * 512);
        if (i < 128) bias_out[i] = i < cout ? bias[i] : 0.f;
    }
}

struct CxParams {
    int b;
    long long L;
    const float *x;                 // (B,C0,L)
    const unsigned char *img;
    float *y;                       // chain3: (B,128,L)
    float *seg, *nocs;              // tail: (B,SEG,L), (B,NOCS,L)
    int act3, nocs_act;
};

template <int C0, int SEG, int NOCS, bool TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void chain_x6_kernel(CxParams p) {
    using S = CxShape<C0, TAIL>;
    constexpr int NS = 3;
    static_assert(S::NCH % NS == 0, "chunks per slice a multiple of the ring's slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *bias_lds = reinterpret_cast<float *>(smem + NS * S::SLOTB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, col = lane & 31;

    const unsigned long long img_addr = reinterpret_cast<unsigned long long>(p.img);
    const cx_i32x4 wsrc = {(int)(unsigned)img_addr, (int)(unsigned)(img_addr >> 32), S::WBYTES, 0x00020000};
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char *)smem);
    const unsigned voff16 = lane * 16;
    // (chunk / group numbers travel as integral constants: the layer table's constexpr walks must fold at compile time -- called on a
    // lambda's run-time parameter they become real loops, 4600 branches and 134 KB of code in the six-layer kernel)
    auto issue_chunk = [&](auto c_c) __attribute__((always_inline)) {          // (asm, buffer form: see sa_x6.hip)
        constexpr int c = decltype(c_c)::value;
        constexpr int slot_off = (c % NS) * S::SLOTB, img_off = S::chunk_off(c), pieces = S::frags(S::layer_of(c)) / 4;
        const unsigned dst = lds0 + slot_off + wave * 1024;
        const unsigned soff = img_off + wave * 1024;
        const unsigned vo = voff16;                         // (copies: an asm operand alone does not capture in a generic lambda)
        const cx_i32x4 ws = wsrc;
#pragma unroll
        for (int i = 0; i < pieces; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst + i * 4096), "v"(vo), "s"(ws), "s"(soff + i * 4096) : "memory");
    };
    auto acquire = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value;
        CX_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        issue_chunk(std::integral_constant<int, (c + 1) % S::NCH>{});
    };

    issue_chunk(std::integral_constant<int, 0>{});
    for (int e = tid; e < S::NBIAS / 4; e += 256)
        reinterpret_cast<uint4 *>(bias_lds)[e] = reinterpret_cast<const uint4 *>(p.img + S::WBYTES)[e];
    CX_WAIT_VM(0);
    __syncthreads();
    issue_chunk(std::integral_constant<int, 1>{});

    const long long spc = (p.L + 31) / 32;                    // slices per cloud
    const long long nslices = (long long)p.b * spc;
    const long long njobs = (nslices + 3) / 4;                // a job = four consecutive slices, one per wave
    u32x4 wr[2][3];
    auto wload = [&](auto gi_c, u32x4 (&dst)[3]) __attribute__((always_inline)) {
        constexpr int gi = decltype(gi_c)::value;
        constexpr int tau = S::tile_of_group(gi), kk = gi - S::group0(tau);
        constexpr int off = (tau % NS) * S::SLOTB + kk * 3072;
        const unsigned char *bp = smem + off + lane * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) dst[s] = *reinterpret_cast<const u32x4 *>(bp + s * 1024);
    };
    wload(std::integral_constant<int, 0>{}, wr[0]);

    // the input column of a slice: channels 16 kk + 8 h .. + 7 of this lane's position, raw
    auto slice_of = [&](long long job_, int &cb_, long long &pos_, bool &live_) {
        const long long sraw = job_ * 4 + wave;
        const long long sl = sraw < nslices ? sraw : nslices - 1;     // (a spare wave recomputes the last slice and stores nothing)
        live_ = sraw < nslices;
        cb_ = (int)(sl / spc);
        pos_ = (sl - (long long)cb_ * spc) * 32;
    };
    auto load_raw = [&](long long job_, float (&raw_)[S::KST0][8]) {
        int cb_; long long pos_; bool live_;
        slice_of(job_, cb_, pos_, live_);
        const long long pc = pos_ + col < p.L ? pos_ + col : p.L - 1;   // clamped column: computed, never stored
        const float *xp = p.x + (size_t)cb_ * C0 * p.L + pc;
#pragma unroll
        for (int kk = 0; kk < S::KST0; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = 16 * kk + 8 * h + e;
                raw_[kk][e] = ch < C0 ? xp[(size_t)ch * p.L] : 0.f;
            }
    };
    float raw[S::KST0][8];
    if ((long long)blockIdx.x < njobs) load_raw(blockIdx.x, raw);

    for (long long job = blockIdx.x; job < njobs; job += gridDim.x) {
        int cb; long long pos; bool live;
        slice_of(job, cb, pos, live);
        const bool col_ok = live && pos + col < p.L;
        const long long jobn = job + gridDim.x;
        const bool has_next = jobn < njobs;
        u32x4 ha[3][S::KST0], hb[3][8];
#pragma unroll
        for (int kk = 0; kk < S::KST0; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned p0, p1, p2;
                cx_split2(raw[kk][2 * i], raw[kk][2 * i + 1], p0, p1, p2);
                ha[0][kk][i] = p0; ha[1][kk][i] = p1; ha[2][kk][i] = p2;
            }
        __builtin_amdgcn_sched_barrier(0);

        // one read-out unit (register pairs 2u, 2u + 1 -> k-step half u >> 2, pair u & 3) of the finished tile (layer l, row tile t)
        auto readout = [&](auto l_c, auto t_c, auto u_c, const f32x16 &acc) {
            constexpr int l = decltype(l_c)::value, t = decltype(t_c)::value, u = decltype(u_c)::value;
            constexpr int jj = u >> 2, i = u & 3;
            if constexpr (S::stores(l)) {
                // (the flipped form -- a lane owns a channel, 16-byte stores along the positions -- was measured slower here: 91 -> 98 us
                // for the three-layer chain; a half-wave's dword store is one 128-byte row segment)
                float *yb = TAIL ? (l == 3 ? p.seg : p.nocs) : p.y;
                constexpr int cout = TAIL ? (l == 3 ? SEG : NOCS) : S::C;
                const int act = TAIL ? (l == 3 ? (int)ACT_NONE : p.nocs_act) : p.act3;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int reg = 8 * jj + 2 * i + r;
                    const int row = 32 * t + 8 * (reg >> 2) + (reg & 3) + 4 * h;
                    if (col_ok && row < cout) yb[((size_t)cb * cout + row) * p.L + pos + col] = apply_act(acc[reg], act);
                }
            } else {
                const float a = relu_bits(acc[8 * jj + 2 * i]), b2 = relu_bits(acc[8 * jj + 2 * i + 1]);
                unsigned p0, p1, p2;
                cx_split2(a, b2, p0, p1, p2);
                if constexpr (S::out_a(l)) { ha[0][2 * t + jj][i] = p0; ha[1][2 * t + jj][i] = p1; ha[2][2 * t + jj][i] = p2; }
                else { hb[0][2 * t + jj][i] = p0; hb[1][2 * t + jj][i] = p1; hb[2][2 * t + jj][i] = p2; }
            }
        };

        f32x16 acc[2], accl[2];
        cx_static_for<0, S::G>([&](auto gi_c) __attribute__((always_inline)) {
            constexpr int gi = decltype(gi_c)::value;
            constexpr int tau = S::tile_of_group(gi);
            constexpr int kk = gi - S::group0(tau);
            constexpr int l = S::layer_of(tau), t = tau - S::tile0(l), kst = S::kst(l);
            if constexpr (kk == kst - 1) acquire(std::integral_constant<int, (tau + 1) % S::NCH>{});          // the next tile's chunk, one group ahead of its first read
            wload(std::integral_constant<int, (gi + 1 < S::G ? gi + 1 : 0)>{}, wr[(gi + 1) & 1]);
            if constexpr (kk == 0) {
                const float4 *bp = reinterpret_cast<const float4 *>(bias_lds + 128 * l + 32 * t + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = bp[2 * q];
                    acc[tau & 1][4 * q + 0] = v.x; acc[tau & 1][4 * q + 1] = v.y; acc[tau & 1][4 * q + 2] = v.z; acc[tau & 1][4 * q + 3] = v.w;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accl[tau & 1][r] = 0.f;
            }
            // ---- the previous tile's read-out, spread over this tile's groups 1 .. kst - 1 ----
            if constexpr (tau > 0) {
                constexpr int lp = S::layer_of(tau - 1), tp = tau - 1 - S::tile0(lp);
                if constexpr (kk == (kst > 1 ? 1 : 0)) acc[(tau - 1) & 1] += accl[(tau - 1) & 1];
                cx_static_for<0, 8>([&](auto u_c) __attribute__((always_inline)) {
                    if constexpr (cx_unit_group(decltype(u_c)::value, kst) == kk)
                        readout(std::integral_constant<int, lp>{}, std::integral_constant<int, tp>{}, u_c, acc[(tau - 1) & 1]);
                });
            }
            // ---- the six products ----
            if constexpr (S::in_a(l)) cx_group(acc[tau & 1], accl[tau & 1], wr[gi & 1], ha[0][kk], ha[1][kk], ha[2][kk]);
            else cx_group(acc[tau & 1], accl[tau & 1], wr[gi & 1], hb[0][kk], hb[1][kk], hb[2][kk]);
            {
                constexpr int nu = (tau > 0) ? ((cx_unit_group(0, kst) == kk) + (cx_unit_group(1, kst) == kk) + (cx_unit_group(2, kst) == kk) + (cx_unit_group(3, kst) == kk) +
                                                 (cx_unit_group(4, kst) == kk) + (cx_unit_group(5, kst) == kk) + (cx_unit_group(6, kst) == kk) + (cx_unit_group(7, kst) == kk)) : 0;
                constexpr int per = (nu * 13 + 5) / 6;
                if constexpr (per > 0) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the slice's last tile: read out in the open
        {
            constexpr int TL = S::NCH - 1, lp = S::layer_of(TL), tp = TL - S::tile0(lp);
            acc[TL & 1] += accl[TL & 1];
            cx_static_for<0, 8>([&](auto u_c) __attribute__((always_inline)) {
                readout(std::integral_constant<int, lp>{}, std::integral_constant<int, tp>{}, u_c, acc[TL & 1]);
            });
        }
        // (requested under the slice's second layer instead, the 72 loads bought nothing -- 91.5 vs 93.1 us -- and their 72 registers
        // spilled the six-layer kernel: the column is fetched here, at the slice's edge)
        if (has_next) load_raw(jobn, raw);
    }
    CX_WAIT_VM(0);                                        // nothing of the ring may land in LDS after the workgroup is gone
}

template <int C0, int SEG, int NOCS, bool TAIL>
int cx_launch(const CxParams &p, hipStream_t stream) {
    using S = CxShape<C0, TAIL>;
    const int lds = 3 * S::SLOTB + S::NBIAS * 4;
    auto kern = chain_x6_kernel<C0, SEG, NOCS, TAIL>;
    static CaptraDeviceOnce once;
    if (once.first_use()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return (int)hipGetLastError();
        once.done();
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    static std::atomic<int> cus_of[128];
    cus = cus_of[dev & 127].load(std::memory_order_relaxed);
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_of[dev & 127].store(cus, std::memory_order_relaxed);
    }
    const long long njobs = ((long long)p.b * ((p.L + 31) / 32) + 3) / 4;
    const unsigned grid = (unsigned)(njobs < cus ? njobs : cus);
    CAPTRA_LAUNCH(TAIL ? "coord_tail_x6" : "mlp_chain3_x6", kern, dim3(grid), dim3(256), lds, stream, p);
    return captra_last_error();
}

}  // namespace

// bytes of the weight image of a chain: layer widths (cin -> cout) in order; fragments of every layer, then 128 fp32 bias slots per layer
extern "C" long long captra_chain_x6_image_bytes(int nl, const int *cin, const int *cout) {
    if (nl < 1 || nl > 6) return -1;
    long long frags = 0;
    for (int l = 0; l < nl; ++l) frags += (long long)((cout[l] + 31) / 32) * ((((cin[l] + 15) / 16) * 3 + 3) / 4 * 4);
    return frags * 1024 + (long long)nl * 128 * 4;
}

// Layer l of a chain into its place of the image: wt / bias = the layer's PACKED fp32 buffers, natural = 1 for the chain's first layer
// (operand loaded in channel order), 0 for the others (hand-over order).  frag_off = bytes of the fragments of the layers before it,
// total_frag_bytes = of all layers (the biases sit behind them).
extern "C" int captra_pack_chain_x6(int l, int cin, int cout, int natural, long long frag_off, long long total_frag_bytes, const float *wt,
                                    const float *bias, unsigned char *img, captra_stream_t stream) {
    if (l < 0 || l > 5 || cin < 1 || cout < 1 || cout > 128 || wt == nullptr || bias == nullptr || img == nullptr) return -1;
    const int kst = (cin + 15) / 16, nt = (cout + 31) / 32, ch = (kst * 3 + 3) / 4 * 4;
    const long long threads = (long long)nt * ch * 512 + 128;
    CAPTRA_LAUNCH("pack_weights", pack_chain_x6_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cin, cout,
                  (cout + 127) / 128 * 128, natural, wt, bias, img + frag_off, reinterpret_cast<float *>(img + total_frag_bytes) + 128 * l);
    return captra_last_error();
}

extern "C" int captra_mlp_chain3_x6(int b, int c0, long long l, const float *x, const unsigned char *img, int act3, float *y,
                                    captra_stream_t stream) {
    if (b < 0 || c0 < 1 || l < 0 || act3 < 0 || act3 > 2 || img == nullptr) return -1;
    if (b == 0 || l == 0) return 0;
    CxParams p = {};
    p.b = b; p.L = l; p.x = x; p.img = img; p.y = y; p.act3 = act3;
    if (c0 == 131) return cx_launch<131, 0, 0, false>(p, (hipStream_t)stream);
    if (c0 == 134) return cx_launch<134, 0, 0, false>(p, (hipStream_t)stream);
    return -2;
}

extern "C" int captra_coord_tail_x6(int b, int c0, int seg_dim, int nocs_dim, long long l, const float *x, const unsigned char *img,
                                    int nocs_act, float *seg, float *nocs, captra_stream_t stream) {
    if (b < 0 || c0 < 1 || seg_dim < 1 || nocs_dim < 1 || l < 0 || nocs_act < 0 || nocs_act > 2 || img == nullptr) return -1;
    if (b == 0 || l == 0) return 0;
    CxParams p = {};
    p.b = b; p.L = l; p.x = x; p.img = img; p.seg = seg; p.nocs = nocs; p.nocs_act = nocs_act;
#define CX_TAIL(C0_, S_, N_) if (c0 == C0_ && seg_dim == S_ && nocs_dim == N_) return cx_launch<C0_, S_, N_, true>(p, (hipStream_t)stream);
    CX_TAIL(134, 2, 3)      // rigid categories: 1 part + background, 3 NOCS channels
    CX_TAIL(134, 4, 12)     // drawers: 4 parts
    CX_TAIL(134, 3, 9)      // glasses: 3 parts
    CX_TAIL(134, 2, 6)      // scissors / laptop: 2 parts
#undef CX_TAIL
    return -2;
}
