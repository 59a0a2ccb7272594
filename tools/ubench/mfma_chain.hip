// Micro-benchmark: fp32 MFMA 32x32x2 issue rate vs number of independent accumulator chains per wave
// and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSB>
__global__ __launch_bounds__(1024) void chain(float *out, int iters, float seed) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed + i * 1e-9f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = seed, b = seed * 0.5f;
    const float *bp = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (LDSB) b = bp[((it * 8 + j) & 31) * 64];
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDSB>
void run(int waves_per_cu) {
    float *out;
    hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
    int iters = 4096 / NACC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), block(waves_per_cu * 64);
    hipLaunchKernelGGL((chain<NACC, LDSB>), grid, block, 0, 0, out, 16, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC, LDSB>), grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * waves_per_cu * iters * 8.0 * NACC * 32 * 32 * 2 * 2;
    printf("nacc=%d ldsB=%d waves/CU=%2d (%d/SIMD): %7.3f ms  %6.1f TFLOP/s\n", NACC, (int)LDSB, waves_per_cu, waves_per_cu / 4, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w : {4, 8, 16}) { run<1, false>(w); run<2, false>(w); run<4, false>(w); }
    for (int w : {4, 8, 16}) { run<1, true>(w); run<2, true>(w); run<4, true>(w); }
    return 0;
}
