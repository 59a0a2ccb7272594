// Micro-benchmark of the fused-SA inner loop in isolation: A operand via prefetched buffer loads,
// B operand from LDS, one accumulator chain per wave, no barriers.  Variants isolate each ingredient.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: A+B regs invariant, 1: B from LDS, 2: A from buffer loads (prefetch), 3: both
__global__ __launch_bounds__(256) void layer(const float *wt, int ldw, int nsets, int reps, float *out) {
    __shared__ float X[352 * 32];
    for (int i = threadIdx.x; i < 352 * 32; i += 256) X[i] = 1e-3f * (i % 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wt, 0, 352 * ldw * 4, 0x00020000);
    const int kstep_bytes = 2 * ldw * 4;
    const int voff = (((lane >> 5) * ldw) + wave * 32 + (lane & 31)) * 4;
    const float *xrow = X + (lane >> 5) * 32 + (lane & 31);
    constexpr int KS = 16;
    const int set_bytes = KS * kstep_bytes;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float s0[KS], s1[KS], bv[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) { s0[j] = 1.0f + j; s1[j] = 2.0f + j; bv[j] = 0.5f; }
#define LOAD_SET(dst, si)                                                                                     \
    if (MODE & 2) {                                                                                           \
        _Pragma("unroll") for (int j = 0; j < KS; ++j) dst[j] = __builtin_bit_cast(                          \
            float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (si) * set_bytes + j * kstep_bytes, 0)); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define MFMA_SET(src, si)                                                                                     \
    {                                                                                                         \
        if (MODE & 1) {                                                                                       \
            const float *xr = xrow + (size_t)(si) * 32 * 32;                                                  \
            _Pragma("unroll") for (int j = 0; j < KS; ++j) bv[j] = xr[j * 2 * 32];                           \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < KS; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(src[j], bv[j], acc, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    for (int rep = 0; rep < reps; ++rep) {
        LOAD_SET(s0, 0)
        for (int c = 0; c + 1 < nsets; c += 2) {
            LOAD_SET(s1, c + 1)
            MFMA_SET(s0, c)
            LOAD_SET(s0, (c + 2 < nsets ? c + 2 : nsets - 1))
            MFMA_SET(s1, c + 1)
        }
        if (nsets & 1) MFMA_SET(s0, nsets - 1)
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks_per_cu, int nsets) {
    float *out, *wt;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    (void)hipMalloc(&wt, 352 * 256 * sizeof(float));
    (void)hipMemset(wt, 0, 352 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 400;
    dim3 grid(256 * blocks_per_cu), block(256);
    hipLaunchKernelGGL(layer<MODE>, grid, block, 0, 0, wt, 256, nsets, 2, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(layer<MODE>, grid, block, 0, 0, wt, 256, nsets, reps, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * blocks_per_cu * 4 * reps * nsets * 16.0 * 4096;
    printf("mode=%d blocks/CU=%d (waves/SIMD=%d) nsets=%2d: %7.3f ms  %6.1f TFLOP/s\n", MODE, blocks_per_cu, blocks_per_cu, nsets, ms, flops / ms / 1e9);
    (void)hipFree(out); (void)hipFree(wt);
}

int main() {
    for (int bpc : {1, 2, 4}) { run<0>(bpc, 11); run<1>(bpc, 11); run<2>(bpc, 11); run<3>(bpc, 11); }
    for (int ns : {1, 2, 3, 4, 7}) run<3>(2, ns);
    return 0;
}
