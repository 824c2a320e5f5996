// micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD, dependent vs independent
// accumulators, operands in VGPRs
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void rate(const float *in, float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[threadIdx.x * 8 + i]; b[i] = (__bf16)in[2048 + threadIdx.x * 8 + i]; }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int run_rate(int nacc, const float *in, float *out, int iters, void *st) {
    dim3 g(256), b(256);
    if (nacc == 1) hipLaunchKernelGGL(rate<1>, g, b, 0, (hipStream_t)st, in, out, iters);
    else if (nacc == 2) hipLaunchKernelGGL(rate<2>, g, b, 0, (hipStream_t)st, in, out, iters);
    else hipLaunchKernelGGL(rate<4>, g, b, 0, (hipStream_t)st, in, out, iters);
    return (int)hipGetLastError();
}
