// What a loop of dependent v_mfma_f32_32x32x2_f32 (and v_mfma_f32_32x32x16_bf16) really sustains on gfx950, by waves per SIMD and by the number
// of independent accumulator chains -- the ceiling the evaluation / InfoNCE tiles are priced against (EXPERIMENTS.md A.8).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int CHAINS, int VALU>
__global__ __launch_bounds__(256) void f32_kernel(float *out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float junk = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32 / CHAINS; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        if (VALU) {      // the tile epilogue's kind of work: a max over the 16 results, a compare, a few selects (VALU x 4 instructions)
#pragma unroll
            for (int v = 0; v < VALU; ++v) {
                float m = acc[0][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[0][i] + (float)v);
                junk += m > 1e30f ? 1.f : 0.f;
            }
        }
    }
    float s = junk;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void bf16_kernel(float *out, int iters, float a0) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + i); b[i] = (__bf16)(a0 * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32 / CHAINS; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float *out; hipMalloc(&out, 4);
    const int iters = 4000;
    printf("[");
    bool first = true;
    for (int wps : {1, 2, 4}) {                    // waves per SIMD: blocks of 4 waves (one per SIMD), wps blocks per CU
        const int blocks = 256 * wps;
        auto report = [&](const char *what, int chains, int valu, double ms, double flop_per_mfma) {
            const double flops = (double)blocks * 4 * iters * 32 * flop_per_mfma;
            printf("%s\n {\"kernel\": \"%s\", \"waves_per_simd\": %d, \"chains\": %d, \"valu_blocks\": %d, \"ms\": %.4f, \"TFLOPs\": %.1f}", first ? "" : ",", what, wps, chains, valu, ms,
                   flops / (ms * 1e-3) / 1e12);
            first = false;
        };
        report("f32_32x32x2", 1, 0, time_ms([&] { hipLaunchKernelGGL((f32_kernel<1, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 4096);
        report("f32_32x32x2", 2, 0, time_ms([&] { hipLaunchKernelGGL((f32_kernel<2, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 4096);
        report("f32_32x32x2", 4, 0, time_ms([&] { hipLaunchKernelGGL((f32_kernel<4, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 4096);
        report("f32_32x32x2", 1, 1, time_ms([&] { hipLaunchKernelGGL((f32_kernel<1, 1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 4096);
        report("f32_32x32x2", 1, 4, time_ms([&] { hipLaunchKernelGGL((f32_kernel<1, 4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 4096);
        report("bf16_32x32x16", 1, 0, time_ms([&] { hipLaunchKernelGGL((bf16_kernel<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }), 32768);
        report("bf16_32x32x16", 2, 0, time_ms([&] { hipLaunchKernelGGL((bf16_kernel<2>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }), 32768);
    }
    printf("\n]\n");
    return 0;
}
