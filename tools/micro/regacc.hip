// Prototype "swept v2": column-swept SpMM with the accumulators in REGISTERS.
// A lane group of the LDS kernel owns only ~10 accumulator slots (160 KiB / 1024 lanes = 10 float4 per lane), so they fit the
// register file.  A per-lane dynamic register index does not exist, but a wave-uniform one does (a uniform switch): the four
// lane groups of a wave work, at every step, on the SAME local slot number k (four different rows: the k-th chunk "quad" of the
// wave), so the step stream carries one k per step and the sweep touches no LDS at all.  LDS is only the staging area of the
// flush (chunk sums of heavy rows + epilogue), exactly as in spmm_swept.hip.
// Stream metadata: ONE coalesced dword load per 16 steps (lane g*16+j holds (step j, group g)) + DPP row broadcasts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KMAX 12
template <int J> __device__ __forceinline__ int bc_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);      // row_newbcast:J
}
template <int J> __device__ __forceinline__ float bc_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xF, 0xF, false));
}
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void spmm_regacc_kernel(const int32_t *__restrict__ pack, const float *__restrict__ val,
                                                           const int32_t *__restrict__ w_start,      // 64-dword blocks
                                                           const int32_t *__restrict__ w_blocks,
                                                           const int32_t *__restrict__ slot_of,      // [n_waves][KMAX][4] LDS slot of (k, g) or -1
                                                           const float *__restrict__ X, float *__restrict__ Y,
                                                           const int32_t *__restrict__ fptr, const int32_t *__restrict__ frow,
                                                           const int32_t *__restrict__ fstart, const int32_t *__restrict__ fn,
                                                           int n_slots) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = lane & 15, g = lane >> 4;
    const int wid = blockIdx.x * 16 + __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = w_blocks[wid];
    const int32_t *pl = pack + (size_t)w_start[wid] * 64 + lane;
    const float *vl = val + (size_t)w_start[wid] * 64 + lane;
    const char *Xb = reinterpret_cast<const char *>(X);
    f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, a8 = a0, a9 = a0, a10 = a0, a11 = a0;
    f4 x0, x1, x2, x3, y0, y1, y2, y3;
#define GATHER(DST, PK) DST = f4{0.f, 0.f, 0.f, 0.f}; \
    if (MODE != 2 && ((PK) & 0xFFFFF) != 0xFFFFF) DST = *reinterpret_cast<const f4 *>(Xb + (size_t)((PK) & 0xFFFFF) * 256 + sub * 16);
#define ACCUM(PK, VV, XX) {                                                                 \
        const int k = __builtin_amdgcn_readfirstlane((PK) >> 20) & 15;                      \
        const f4 t = XX * VV;                                                               \
        switch (k) {                                                                        \
            case 0: a0 += t; break; case 1: a1 += t; break; case 2: a2 += t; break; case 3: a3 += t; break;   \
            case 4: a4 += t; break; case 5: a5 += t; break; case 6: a6 += t; break; case 7: a7 += t; break;   \
            case 8: a8 += t; break; case 9: a9 += t; break; case 10: a10 += t; break; default: a11 += t; break; \
        } }
#define G4(PV, O, P) GATHER(P##0, bc_i<O + 0>(PV)) GATHER(P##1, bc_i<O + 1>(PV)) GATHER(P##2, bc_i<O + 2>(PV)) GATHER(P##3, bc_i<O + 3>(PV))
#define A4(PV, VV, O, P) ACCUM(bc_i<O + 0>(PV), bc_f<O + 0>(VV), P##0) ACCUM(bc_i<O + 1>(PV), bc_f<O + 1>(VV), P##1)   \
                         ACCUM(bc_i<O + 2>(PV), bc_f<O + 2>(VV), P##2) ACCUM(bc_i<O + 3>(PV), bc_f<O + 3>(VV), P##3)
    if (nblk > 0) {
        int pv = pl[0];
        float vv = vl[0];
        G4(pv, 0, x)
        for (int b = 0; b < nblk; ++b) {
            int pn = -1;                    // an all-pad block: no gathers
            float vn = 0.f;
            if (b + 1 < nblk) { pn = pl[(size_t)(b + 1) * 64]; vn = vl[(size_t)(b + 1) * 64]; }
            G4(pv, 4, y)
            A4(pv, vv, 0, x)
            G4(pv, 8, x)
            A4(pv, vv, 4, y)
            G4(pv, 12, y)
            A4(pv, vv, 8, x)
            G4(pn, 0, x)
            A4(pv, vv, 12, y)
            pv = pn; vv = vn;
        }
    }
    // registers -> LDS slots (each (wave, k, group) owns one slot or none), then the flush of spmm_swept.hip
    const int32_t *so = slot_of + (size_t)wid * KMAX * 4 + g;
#define PUT(K, A) { const int s = so[K * 4]; if (s >= 0) lds[s * 16 + sub] = make_float4(A[0], A[1], A[2], A[3]); }
    PUT(0, a0) PUT(1, a1) PUT(2, a2) PUT(3, a3) PUT(4, a4) PUT(5, a5) PUT(6, a6) PUT(7, a7) PUT(8, a8) PUT(9, a9) PUT(10, a10) PUT(11, a11)
    __syncthreads();
    float4 *Y4 = reinterpret_cast<float4 *>(Y);
    const int f0 = fptr[blockIdx.x], f1 = fptr[blockIdx.x + 1];
    const int rl = tid / 16, rs = tid % 16;
    for (int i = f0 + rl; i < f1; i += 64) {
        const int s0 = fstart[i], n = fn[i];
        float4 t = lds[s0 * 16 + rs];
        for (int k = 1; k < n; ++k) {
            const float4 u = lds[(s0 + k) * 16 + rs];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        Y4[(size_t)frow[i] * 16 + rs] = t;
    }
}

extern "C" int launch_regacc(const int32_t *pack, const float *val, const int32_t *w_start, const int32_t *w_blocks,
                             const int32_t *slot_of, const float *X, float *Y, const int32_t *fptr, const int32_t *frow,
                             const int32_t *fstart, const int32_t *fn, int n_slots, int n_blocks, int mode, void *stream) {
    const size_t ldsb = (size_t)n_slots * 256;
#define GO(M) { hipFuncSetAttribute((const void *)spmm_regacc_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
    hipLaunchKernelGGL((spmm_regacc_kernel<M>), dim3(n_blocks), dim3(1024), ldsb, (hipStream_t)stream, pack, val, w_start, w_blocks, \
                       slot_of, X, Y, fptr, frow, fstart, fn, n_slots); }
    if (mode == 0) GO(0) else GO(2)
    return (int)hipGetLastError();
}
