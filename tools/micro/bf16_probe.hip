// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950 (assumed: lane l supplies
// A[m = l & 31][k = (l >> 5) * 8 + i] and B[k = (l >> 5) * 8 + i][n = l & 31], i = 0..7).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float *A, const float *B, float *C) {   // A[32][16], B[16][32], C[32][32]
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)A[r * 16 + h * 8 + i];
        b[i] = (__bf16)B[(h * 8 + i) * 32 + r];
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = c[i];
    }
}
extern "C" int run_probe(const float *A, const float *B, float *C, void *stream) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C);
    return (int)hipGetLastError();
}
