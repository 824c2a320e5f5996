// Counter-based random numbers for the device-side augmentations: Philox4x32-10 (Salmon et al.,
// SC'11), the generator family PyTorch's device RNG uses.  The reference draws EdgeDrop masks and EmbedPerturb noise
// with `t.rand` on the CPU generator and ships them over PCIe (models/aug_utils.py:28,130); in perf mode
// (model.device_rng) nothing is drawn, stored or copied: a kernel computes the uniform it needs from
//     key     = seed                                (philox_state[0], device memory)
//     counter = (element index lo, hi, stream, step) (step = philox_state[1], advanced once per training step by
//                                                    sslrec_philox_advance -- device memory, so a captured hipGraph
//                                                    draws fresh numbers on every replay; stream = a per-call constant)
// so the SAME element of the SAME call always sees the SAME number (forward and backward views agree), statistically
// equivalent to the reference's draws, not bit-equal.  u = (x >> 8) * 2^-24 in [0, 1), like torch.rand's fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PhiloxKey { uint32_t k0, k1, step; };

__device__ __forceinline__ PhiloxKey philox_load(const uint64_t *state) {
    const uint64_t seed = state[0], step = state[1];
    return PhiloxKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step};
}

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float philox_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }      // 2^-24

// four uniforms for element group `index` (64-bit) of call `stream`
__device__ __forceinline__ float4 philox_uniform4(const PhiloxKey &k, uint64_t index, uint32_t stream) {
    const uint4 r = philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), stream, k.step, k.k0, k.k1);
    return make_float4(philox_u01(r.x), philox_u01(r.y), philox_u01(r.z), philox_u01(r.w));
}

// one uniform for element `index` (entries of an EdgeDrop mask: element = index / 4, lane = index % 4)
__device__ __forceinline__ float philox_uniform1(const PhiloxKey &k, uint64_t index, uint32_t stream) {
    const uint4 r = philox4x32_10((uint32_t)(index >> 2), (uint32_t)(index >> 34), stream, k.step, k.k0, k.k1);
    const uint32_t x = (index & 3) == 0 ? r.x : (index & 3) == 1 ? r.y : (index & 3) == 2 ? r.z : r.w;
    return philox_u01(x);
}
