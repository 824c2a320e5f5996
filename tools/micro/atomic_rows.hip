// micro-benchmark (not product): throughput of wave-wide fp32 atomic adds of one 256-byte row into a
// 37 MB table from all XCDs (what a column-sliced SpMM with partial sums through memory would do).
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void atomic_rows(float *Y, const int *rows, int per_wave, int mode) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int *r = rows + (size_t)w * per_wave;
    for (int i = 0; i < per_wave; ++i) {
        float *p = Y + (size_t)r[i] * 64 + lane;
        if (mode == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (mode == 1) unsafeAtomicAdd(p, 1.0f);
        else *p = 1.0f;   // plain store for comparison
    }
}
extern "C" int launch(float *Y, const int *rows, int n_waves, int per_wave, int mode, void *stream) {
    hipLaunchKernelGGL(atomic_rows, dim3(n_waves / 4), dim3(256), 0, (hipStream_t)stream, Y, rows, per_wave, mode);
    return (int)hipGetLastError();
}
