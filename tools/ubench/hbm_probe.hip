// What can this box's HBM sustain?  Pure write (plain / non-temporal 16-byte stores), pure read, copy.
//   hbm_probe [MiB]     default 2048 MiB per buffer
// Used to put the group_points numbers (a pure-write op: 4*C*M*K output bytes, gathers served by LDS) in context.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void fill_kernel(f32x4 *dst, size_t n4, float v) {
    const f32x4 vv = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(vv, dst + i);
        else dst[i] = vv;
    }
}

// each block owns a contiguous chunk (what group_points does) instead of a grid-strided sweep
template <int NT>
__global__ __launch_bounds__(256) void fill_chunk_kernel(f32x4 *dst, size_t n4, size_t chunk4, float v) {
    const f32x4 vv = {v, v, v, v};
    const size_t b0 = (size_t)blockIdx.x * chunk4;
    const size_t b1 = b0 + chunk4 < n4 ? b0 + chunk4 : n4;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
        if (NT) __builtin_nontemporal_store(vv, dst + i);
        else dst[i] = vv;
    }
}

__global__ __launch_bounds__(256) void read_kernel(const f32x4 *src, size_t n4, float *sink) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.f;
}

template <int NT>
__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *src, f32x4 *dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoll(argv[1]) : 2048;
    const size_t bytes = mib << 20, n4 = bytes / 16;
    f32x4 *a, *b;
    float *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    const int reps = 10;
    for (int grid : {2048, 8192, 65536}) {
        double t;
        t = time_ms([&] { fill_kernel<0><<<grid, 256>>>(a, n4, 1.f); }, reps);
        printf("fill  plain  grid %6d: %7.3f ms  %6.2f TB/s\n", grid, t, bytes / t * 1e-9);
        t = time_ms([&] { fill_kernel<1><<<grid, 256>>>(a, n4, 1.f); }, reps);
        printf("fill  nt     grid %6d: %7.3f ms  %6.2f TB/s\n", grid, t, bytes / t * 1e-9);
        t = time_ms([&] { read_kernel<<<grid, 256>>>(a, n4, sink); }, reps);
        printf("read         grid %6d: %7.3f ms  %6.2f TB/s\n", grid, t, bytes / t * 1e-9);
        t = time_ms([&] { copy_kernel<0><<<grid, 256>>>(a, b, n4); }, reps);
        printf("copy  plain  grid %6d: %7.3f ms  %6.2f TB/s (read+write)\n", grid, t, 2.0 * bytes / t * 1e-9);
        t = time_ms([&] { copy_kernel<1><<<grid, 256>>>(a, b, n4); }, reps);
        printf("copy  nt     grid %6d: %7.3f ms  %6.2f TB/s (read+write)\n", grid, t, 2.0 * bytes / t * 1e-9);
    }
    for (size_t chunk_kb : {16, 64, 256, 1024}) {
        const size_t chunk4 = chunk_kb * 1024 / 16;
        const int grid = (int)((n4 + chunk4 - 1) / chunk4);
        double t = time_ms([&] { fill_chunk_kernel<1><<<grid, 256>>>(a, n4, chunk4, 1.f); }, reps);
        printf("fill  nt  chunk %5zu KiB (grid %7d): %7.3f ms  %6.2f TB/s\n", chunk_kb, grid, t, bytes / t * 1e-9);
        t = time_ms([&] { fill_chunk_kernel<0><<<grid, 256>>>(a, n4, chunk4, 1.f); }, reps);
        printf("fill  pl  chunk %5zu KiB (grid %7d): %7.3f ms  %6.2f TB/s\n", chunk_kb, grid, t, bytes / t * 1e-9);
    }
    double t = time_ms([&] { hipMemsetAsync(a, 0, bytes, 0); }, reps);
    printf("hipMemsetAsync: %7.3f ms  %6.2f TB/s\n", t, bytes / t * 1e-9);
    t = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, reps);
    printf("hipMemcpyAsync D2D: %7.3f ms  %6.2f TB/s (read+write)\n", t, 2.0 * bytes / t * 1e-9);
    return 0;
}
