// Ceiling of the vector-memory path for random ROW gathers on gfx950: every wave instruction is a 16-byte-per-lane
// load whose 64 lanes fetch G = 1024 / row_bytes different rows of a [T, row_bytes] table (row ids from an in-register
// LCG: no index stream), K loads in flight per wave, results summed in registers.  This is the data movement of an
// SpMM gather with everything else removed: time(nnz gathers) >= nnz * row_bytes / ceiling(T).
#include <hip/hip_runtime.h>
#include <stdint.h>

// xcd_tables != 0: workgroup b gathers from table b % 8 (workgroups are dealt round-robin to the 8 XCDs, so every XCD
// then has a table of its own: the "feature-sliced" SpMM, where XCD k owns d/8 columns of ALL rows)
template <int LPG, int K>
__global__ __launch_bounds__(1024) void gather_kernel(const float4 *__restrict__ X, unsigned T, int iters, float4 *out, int xcd_tables) {
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPG;
    if (xcd_tables) X += (size_t)(blockIdx.x & 7) * T * LPG;
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
    if (acc.x == 1.2345e-30f) out[threadIdx.x] = acc;
}

extern "C" int launch_gather(const void *X, unsigned T, int row_bytes, int iters, int blocks, int threads, int k, void *out,
                             void *stream, int xcd_tables) {
    const float4 *x = (const float4 *)X;
    float4 *o = (float4 *)out;
    hipStream_t st = (hipStream_t)stream;
#define GO(LPG, K) hipLaunchKernelGGL((gather_kernel<LPG, K>), dim3(blocks), dim3(threads), 0, st, x, T, iters, o, xcd_tables)
    if (row_bytes == 256 && k == 8) GO(16, 8);
    else if (row_bytes == 256 && k == 4) GO(16, 4);
    else if (row_bytes == 128 && k == 8) GO(8, 8);
    else if (row_bytes == 64 && k == 8) GO(4, 8);
    else if (row_bytes == 32 && k == 8) GO(2, 8);
    else if (row_bytes == 32 && k == 16) GO(2, 16);
    else if (row_bytes == 512 && k == 8) GO(32, 8);
    else if (row_bytes == 1024 && k == 8) GO(64, 8);
    else return 1;
    return (int)hipGetLastError();
}
