// How fast does the chip take 256-byte ROW writes, by the order the workgroups write them in?  (the flush of the column-swept SpMM:
// 256 workgroups x 16 waves, a wave instruction writes 4 rows of 256 B)  mode 0: workgroup b writes rows perm[b * R + i] (its rows are
// scattered over the table: the LPT dealing of rows to workgroups); mode 1: workgroup b writes the contiguous range [b * R, (b + 1) * R)
// in a scattered order inside it (contiguous ranges, waves own scattered rows of it); mode 2: the same range front to back.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void row_write_kernel(f4 *out, const int *perm, int rows_per_wg, int mode, int write_through) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rl = lane >> 4, rs = lane & 15;
    const int base = blockIdx.x * rows_per_wg;
    for (int i = wave * 4 + rl; i < rows_per_wg; i += 64) {
        int row;
        if (mode == 0) row = perm[base + i];
        else if (mode == 1) row = base + perm[base + i] % rows_per_wg;
        else row = base + i;
        f4 v = {1.f * row, 2.f, 3.f, 4.f};
        f4 *p = out + (size_t)row * 16 + rs;
        if (write_through) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else *p = v;
    }
}
extern "C" int launch_row_writes(void *out, const int *perm, int n_wg, int rows_per_wg, int mode, int wt, void *stream) {
    hipLaunchKernelGGL(row_write_kernel, dim3(n_wg), dim3(1024), 0, (hipStream_t)stream, (f4 *)out, perm, rows_per_wg, mode, wt);
    return (int)hipGetLastError();
}
