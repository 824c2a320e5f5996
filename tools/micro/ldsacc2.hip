// Prototype v2: LDS-accumulator SpMM, stream metadata through ONE coalesced dword load per 16 steps and
// DPP row broadcasts (row_newbcast) instead of per-step 16-byte loads.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int J> __device__ __forceinline__ int bc_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);      // row_newbcast:J
}
template <int J> __device__ __forceinline__ float bc_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xF, 0xF, false));
}

template <int D, int MODE, int NW>
__global__ __launch_bounds__(1024) void spmm_ldsacc2_kernel(const int32_t *__restrict__ pack, const float *__restrict__ val,
                                                            const int32_t *__restrict__ w_start,
                                                            const int32_t *__restrict__ w_steps,
                                                            const float *__restrict__ X, float *__restrict__ Y,
                                                            const int32_t *__restrict__ fptr,
                                                            const int32_t *__restrict__ frow,
                                                            const int32_t *__restrict__ fstart,
                                                            const int32_t *__restrict__ fn, int n_slots,
                                                            const int32_t *__restrict__ phase_row, int n_phase, int lead) {
    extern __shared__ float4 acc[];
    __shared__ int progress;
    __shared__ float4 dump[64];
    constexpr int G = 256 / D;
    constexpr int LPG = 64 / G;
    constexpr int RV = D / 4;
    static_assert(LPG >= 16, "row broadcast needs >= 16 lanes per output row");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub = lane % LPG;
    for (int i = tid; i < n_slots * RV; i += 1024) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) progress = 0;
    __syncthreads();
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const bool is_pf = (NW < 16 && uw >= NW);
    const int wid = blockIdx.x * NW + (is_pf ? 0 : uw);
    const int base = w_start[wid];           // in 64-dword blocks
    const int nblk = is_pf ? 0 : (w_steps[wid] >> 4);      // 16 steps per block
    const int32_t *pl = pack + (size_t)base * 64 + lane;
    const float *vl = val + (size_t)base * 64 + lane;
    const float4 *X4 = reinterpret_cast<const float4 *>(X);
    float4 dbg = make_float4(0.f, 0.f, 0.f, 0.f);
#define GATHER(PK) (MODE == 2 ? make_float4(1.f, 2.f, 3.f, 4.f) : X4[(size_t)((PK) != -1 ? ((PK) & 0xFFFFF) : 0) * RV + sub])
#define ACCUM(PK, VV, XX)                                                  \
    if (MODE == 1) {                                                       \
        dbg.x = fmaf(VV, XX.x, dbg.x); dbg.y = fmaf(VV, XX.y, dbg.y);      \
        dbg.z = fmaf(VV, XX.z, dbg.z); dbg.w = fmaf(VV, XX.w, dbg.w);      \
    } else if ((PK) != -1) {                                               \
        const int s = (int)((unsigned)(PK) >> 20) * RV + sub;              \
        float4 a = acc[s];                                                 \
        a.x = fmaf(VV, XX.x, a.x); a.y = fmaf(VV, XX.y, a.y);              \
        a.z = fmaf(VV, XX.z, a.z); a.w = fmaf(VV, XX.w, a.w);              \
        acc[s] = a;                                                        \
    }
#define G8(PV, O, X0, X1, X2, X3, X4_, X5, X6, X7)                                                       \
    X0 = GATHER(bc_i<O + 0>(PV)); X1 = GATHER(bc_i<O + 1>(PV)); X2 = GATHER(bc_i<O + 2>(PV));             \
    X3 = GATHER(bc_i<O + 3>(PV)); X4_ = GATHER(bc_i<O + 4>(PV)); X5 = GATHER(bc_i<O + 5>(PV));            \
    X6 = GATHER(bc_i<O + 6>(PV)); X7 = GATHER(bc_i<O + 7>(PV));
#define A8(PV, VV, O, X0, X1, X2, X3, X4_, X5, X6, X7)                                                   \
    ACCUM(bc_i<O + 0>(PV), bc_f<O + 0>(VV), X0) ACCUM(bc_i<O + 1>(PV), bc_f<O + 1>(VV), X1)               \
    ACCUM(bc_i<O + 2>(PV), bc_f<O + 2>(VV), X2) ACCUM(bc_i<O + 3>(PV), bc_f<O + 3>(VV), X3)               \
    ACCUM(bc_i<O + 4>(PV), bc_f<O + 4>(VV), X4_) ACCUM(bc_i<O + 5>(PV), bc_f<O + 5>(VV), X5)              \
    ACCUM(bc_i<O + 6>(PV), bc_f<O + 6>(VV), X6) ACCUM(bc_i<O + 7>(PV), bc_f<O + 7>(VV), X7)
    if (nblk > 0) {
        int pv = pl[0];
        float vv = vl[0];
        float4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
        G8(pv, 0, a0, a1, a2, a3, a4, a5, a6, a7)
        for (int b = 0; b < nblk; ++b) {
            int pn = -1;
            float vn = 0.f;
            if (b + 1 < nblk) {
                pn = pl[(size_t)(b + 1) * 64];
                vn = vl[(size_t)(b + 1) * 64];
            }
            G8(pv, 8, b0, b1, b2, b3, b4, b5, b6, b7)
            if (NW < 16 && wave == 0 && lane == 0) *(volatile int *)&progress = b;
            A8(pv, vv, 0, a0, a1, a2, a3, a4, a5, a6, a7)
            if (b + 1 < nblk) { G8(pn, 0, a0, a1, a2, a3, a4, a5, a6, a7) }
            A8(pv, vv, 8, b0, b1, b2, b3, b4, b5, b6, b7)
            pv = pn;
            vv = vn;
        }
    }
    if (NW < 16 && uw == 0 && lane == 0) *(volatile int *)&progress = 1 << 30;      // sweep finished: release the prefetcher
    if (is_pf && lead >= 0) {
        // dedicated prefetch wave: pulls the X rows of phase p into this XCD's L2 with coalesced 1-KiB loads,
        // `lead` phases ahead of the block's own sweep position (32 blocks per XCD share a phase's rows)
        const int lb = blockIdx.x >> 3;
        const float4 *X4p = reinterpret_cast<const float4 *>(X);
        for (int p = 0; p < n_phase; ++p) {
            for (int spin = 0; spin < (1 << 14) && *(volatile int *)&progress + lead < p; ++spin) __builtin_amdgcn_s_sleep(8);
            const int r0 = phase_row[p], r1 = phase_row[p + 1];
            const int units = (r1 - r0 + 3) >> 2;
            const float4 *src = X4p + (size_t)r0 * RV + lane;
            const int last = (r1 - r0) * RV - 1 - lane;
            for (int u = lb; u < units; u += 32)          // fire and forget: LDS-DMA into a 1-KiB dump area, no VGPRs
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + min(u * 64, last)),
                                                 (__attribute__((address_space(3))) void *)dump, 16, 0, 0);
        }
    }
    if (dbg.x == 1.2345e-30f) dbg.y += dump[lane].x;      // keeps the dump area allocated
    if (MODE == 1 || dbg.x == 1.2345e-30f) acc[tid] = dbg;
    __syncthreads();
    float4 *Y4 = reinterpret_cast<float4 *>(Y);
    const int f0 = fptr[blockIdx.x], f1 = fptr[blockIdx.x + 1];
    const int rl = tid / RV, rs = tid % RV;
    for (int i = f0 + rl; i < f1; i += 1024 / RV) {
        const int s0 = fstart[i], n = fn[i];
        float4 t = acc[s0 * RV + rs];
        for (int k = 1; k < n; ++k) {
            const float4 u = acc[(s0 + k) * RV + rs];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        Y4[(size_t)frow[i] * RV + rs] = t;
    }
}

template <int D, int MODE, int NW>
static int launch_t(const int32_t *pack, const float *val, const int32_t *w_start, const int32_t *w_steps, const float *X,
                    float *Y, const int32_t *fptr, const int32_t *frow, const int32_t *fstart, const int32_t *fn,
                    int n_slots, int n_blocks, const int32_t *phase_row, int n_phase, int lead, void *stream) {
    const size_t lds = (size_t)n_slots * D * 4;
    hipFuncSetAttribute((const void *)spmm_ldsacc2_kernel<D, MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((spmm_ldsacc2_kernel<D, MODE, NW>), dim3(n_blocks), dim3(1024), lds, (hipStream_t)stream, pack, val,
                       w_start, w_steps, X, Y, fptr, frow, fstart, fn, n_slots, phase_row, n_phase, lead);
    return (int)hipGetLastError();
}

extern "C" int launch_ldsacc2(const int32_t *pack, const float *val, const int32_t *w_start, const int32_t *w_steps,
                              const float *X, float *Y, const int32_t *fptr, const int32_t *frow, const int32_t *fstart,
                              const int32_t *fn, int n_slots, int n_blocks, int d, int mode,
                              const int32_t *phase_row, int n_phase, int lead, void *stream) {
#define ARGS pack, val, w_start, w_steps, X, Y, fptr, frow, fstart, fn, n_slots, n_blocks, phase_row, n_phase, lead, stream
    if (d == 64 && lead < 0) return mode == 0 ? launch_t<64, 0, 16>(ARGS) : mode == 1 ? launch_t<64, 1, 16>(ARGS) : launch_t<64, 2, 16>(ARGS);
    if (d == 64) return mode == 0 ? launch_t<64, 0, 15>(ARGS) : mode == 1 ? launch_t<64, 1, 15>(ARGS) : launch_t<64, 2, 15>(ARGS);
    return 1;
}
