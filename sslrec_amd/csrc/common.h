// Shared device helpers for libsslrec_hip (gfx950 only; wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sslrec_hip.h"

#define SSLREC_WAVE 64

#define SSLREC_LAUNCH_CHECK()                      \
    do {                                           \
        hipError_t _e = hipGetLastError();         \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

// sum over the 64 lanes of a wavefront; every lane receives the total
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wave index inside the workgroup as a provably wave-uniform (SGPR) value
__device__ __forceinline__ int wave_in_block() {
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

__device__ __forceinline__ float sign_f(float x) {
    return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
}

// ---- in-kernel launch timing (sslrec_debug_stamp_next_launch, include/sslrec_hip.h) ------------------------------------
// record = {min start clock (init ~0), sum of durations, arrivals, launches}
__device__ __forceinline__ void stamp_begin(unsigned long long *rec) {
    if (rec && threadIdx.x == 0) atomicMin(&rec[0], (unsigned long long)wall_clock64());
}
// BARRIER = true: call from EVERY thread of the workgroup at the very end (waits for all of its waves);
// BARRIER = false: the workgroup's first wave speaks for it (kernels whose waves may have exited: equal-length streams)
template <bool BARRIER>
__device__ __forceinline__ void stamp_end(unsigned long long *rec) {
    if (!rec) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have left the CU
    if constexpr (BARRIER) __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long now = (unsigned long long)wall_clock64();
        const unsigned long long arrived = atomicAdd(&rec[2], 1ull);
        if (arrived + 1 == (unsigned long long)gridDim.x) {
            const unsigned long long t0 = atomicExch(&rec[0], ~0ull);
            atomicExch(&rec[2], 0ull);
            atomicAdd(&rec[1], now - t0);
            atomicAdd(&rec[3], 1ull);
        }
    }
}
// host side: the record the next SpMM launch of this thread takes (consumed by the launch)
unsigned long long *sslrec_take_stamp();
