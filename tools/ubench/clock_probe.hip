// What do s_memtime / s_memrealtime tick at, and what is the shader clock under load?
//   clock_probe <mode>   mode 0: idle chip, one wave;  mode 1: every SIMD runs an fp32-MFMA chain meanwhile
// Each probed wave runs a dependent chain of N v_add_f32 (4 cycles each, 1 wave/SIMD: no contention) or N MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(unsigned long long *out, int n, int use_mfma) {
    float v = threadIdx.x;
    f32x16 acc = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (use_mfma) {
        for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 1.0f, acc, 0, 0, 0);
    } else {
        for (int i = 0; i < n; ++i) v = v * 1.0000001f + 1.0f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = r1 - r0;
    }
    if (v == 12345.f || acc[0] == 12345.f) out[0] = 0;
}

int main(int argc, char **argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 1;     // 1 = idle chip; 1024 = one wave per SIMD
    int use_mfma = argc > 2 ? atoi(argv[2]) : 0;
    int n = 200000;
    unsigned long long *d, h[4096];
    hipMalloc(&d, sizeof(h));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<<<blocks, 64>>>(d, n, use_mfma);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<<<blocks, 64>>>(d, n, use_mfma);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double per = use_mfma ? 64.0 : 4.0;   // issue cycles of one dependent instruction (MFMA 32x32x2: 16 passes x 4; VALU: 4)
    printf("blocks %d mfma %d: kernel %.1f us; wave0 s_memtime ticks %llu, s_memrealtime ticks %llu -> memtime/realtime = %.3f; "
           "chain of %d ops (>= %.0f cycles) => shader clock >= %.3f GHz if realtime is 100 MHz; memtime rate %.3f GHz\n",
           blocks, use_mfma, ms * 1e3, h[0], h[1], (double)h[0] / h[1], n, n * per, n * per / (h[1] / 100e6) / 1e9,
           (double)h[0] / (h[1] / 100e6) / 1e9);
    return 0;
}
