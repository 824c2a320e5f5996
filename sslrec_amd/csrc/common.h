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
