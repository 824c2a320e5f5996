// Measurement aid, not product code: the ceiling of the vector-memory path for random ROW gathers on gfx950 -- the data movement of
// an SpMM gather with everything else removed.  bench.py runs it in the same process, on the same table size, right beside the SpMM
// it prices (`roofline.ceiling`): a one-gather-per-edge kernel needs at least nnz * row_bytes / (this rate) per launch, whatever its
// layout, so `roofline.frac` can be read against what ANY such kernel could reach on this graph.
//
// Every wave instruction is a 16-byte-per-lane load whose 64 lanes fetch 1024 / row_bytes different rows of a [T, row_bytes] table
// (row ids from an in-register LCG: no index stream), 8 loads in flight per wave, results summed in registers.
#include "common.h"

template <int LPG, int K>
__global__ __launch_bounds__(1024) void gather_rows_kernel(const float4 *__restrict__ X, unsigned T, int iters, float4 *out) {
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPG;
    const unsigned gid = (blockIdx.x * blockDim.x + threadIdx.x) / LPG;       // lane group id
    unsigned s = gid * 2654435761u + 12345u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = (unsigned)(((unsigned long long)(s >> 4) * T) >> 28);
            v[k] = X[(size_t)r * LPG + sub];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    if (acc.x == 1.2345e-30f) out[threadIdx.x] = acc;      // (never true: keeps the loads alive)
}

extern "C" int sslrec_debug_gather_rows(const float *X, uint32_t n_rows, int32_t row_bytes, int32_t iters, int32_t blocks, float *scratch,
                                        int64_t *n_gathers, void *stream) {
    if (!X || !scratch || n_rows == 0 || iters < 1 || blocks < 1) return SSLREC_E_BADARG;
    const float4 *x = reinterpret_cast<const float4 *>(X);
    float4 *o = reinterpret_cast<float4 *>(scratch);              // >= 1024 float4
    hipStream_t st = (hipStream_t)stream;
    int lpg;
    if (row_bytes == 128) { lpg = 8; hipLaunchKernelGGL((gather_rows_kernel<8, 8>), dim3(blocks), dim3(1024), 0, st, x, n_rows, iters, o); }
    else if (row_bytes == 256) { lpg = 16; hipLaunchKernelGGL((gather_rows_kernel<16, 8>), dim3(blocks), dim3(1024), 0, st, x, n_rows, iters, o); }
    else if (row_bytes == 512) { lpg = 32; hipLaunchKernelGGL((gather_rows_kernel<32, 8>), dim3(blocks), dim3(1024), 0, st, x, n_rows, iters, o); }
    else return SSLREC_E_BADARG;
    SSLREC_LAUNCH_CHECK();
    if (n_gathers) *n_gathers = (int64_t)blocks * 1024 / lpg * iters * 8;
    return 0;
}
