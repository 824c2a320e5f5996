// Prototype (micro-benchmark, not part of libsslrec_hip): SpMM with the OUTPUT rows accumulated in LDS.
// One workgroup per CU owns <= SLOTS output rows (all of LDS); its 16 waves walk edge streams sorted by
// COLUMN, so that the 32 CUs of an XCD sweep the X table together and every X row crosses the fabric
// about once per XCD instead of once per L2 miss.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int D, int MODE>
__global__ __launch_bounds__(1024) void spmm_ldsacc_kernel(const int32_t *__restrict__ pack, const float *__restrict__ val,
                                                           const int32_t *__restrict__ w_start,
                                                           const int32_t *__restrict__ w_steps,
                                                           const float *__restrict__ X, float *__restrict__ Y,
                                                           const int32_t *__restrict__ fptr,
                                                           const int32_t *__restrict__ frow,
                                                           const int32_t *__restrict__ fstart,
                                                           const int32_t *__restrict__ fn, int n_slots, int sync_every) {
    extern __shared__ float4 acc[];
    __shared__ int nblk_max;
    constexpr int G = 256 / D;        // output rows per wave instruction
    constexpr int LPG = 64 / G;       // lanes per row (each lane owns a float4)
    constexpr int RV = D / 4;         // float4 per row
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub = lane % LPG, g = lane / LPG;
    for (int i = tid; i < n_slots * RV; i += 1024) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) nblk_max = 0;
    __syncthreads();
    const int wid = blockIdx.x * 16 + __builtin_amdgcn_readfirstlane(wave);
    const int base = w_start[wid];
    const int nblk = w_steps[wid] >> 2;
    const int4 *pp = reinterpret_cast<const int4 *>(pack + base) + g;
    const float4 *vp = reinterpret_cast<const float4 *>(val + base) + g;
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
    if (sync_every > 0) {
        if (lane == 0) atomicMax(&nblk_max, nblk);
        __syncthreads();
        const int nb = nblk_max;
        int4 pl = nblk > 0 ? pp[0] : make_int4(-1, -1, -1, -1);
        float4 vl = nblk > 0 ? vp[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        int p0 = pl.x, p1 = pl.y, p2 = pl.z, p3 = pl.w;
        float v0 = vl.x, v1 = vl.y, v2 = vl.z, v3 = vl.w;
        float4 x0 = GATHER(p0), x1 = GATHER(p1), x2 = GATHER(p2), x3 = GATHER(p3);
        for (int b = 1; b <= nb; ++b) {
            if (b % sync_every == 0) __syncthreads();
            int q0 = -1, q1 = -1, q2 = -1, q3 = -1;
            if (b < nblk) {
                pl = pp[b * G];
                vl = vp[b * G];
                q0 = pl.x; q1 = pl.y; q2 = pl.z; q3 = pl.w;
            }
            const float4 y0 = GATHER(q0), y1 = GATHER(q1), y2 = GATHER(q2), y3 = GATHER(q3);
            ACCUM(p0, v0, x0) ACCUM(p1, v1, x1) ACCUM(p2, v2, x2) ACCUM(p3, v3, x3)
            p0 = q0; p1 = q1; p2 = q2; p3 = q3;
            v0 = vl.x; v1 = vl.y; v2 = vl.z; v3 = vl.w;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        }
    } else if (nblk > 0) {
        int4 pl = pp[0];
        float4 vl = vp[0];
        int p0 = pl.x, p1 = pl.y, p2 = pl.z, p3 = pl.w;
        float v0 = vl.x, v1 = vl.y, v2 = vl.z, v3 = vl.w;
        float4 x0 = GATHER(p0), x1 = GATHER(p1), x2 = GATHER(p2), x3 = GATHER(p3);
        for (int b = 1; b < nblk; ++b) {
            pl = pp[b * G];
            vl = vp[b * G];
            const int q0 = pl.x, q1 = pl.y, q2 = pl.z, q3 = pl.w;
            const float4 y0 = GATHER(q0), y1 = GATHER(q1), y2 = GATHER(q2), y3 = GATHER(q3);
            ACCUM(p0, v0, x0) ACCUM(p1, v1, x1) ACCUM(p2, v2, x2) ACCUM(p3, v3, x3)
            p0 = q0; p1 = q1; p2 = q2; p3 = q3;
            v0 = vl.x; v1 = vl.y; v2 = vl.z; v3 = vl.w;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        }
        ACCUM(p0, v0, x0) ACCUM(p1, v1, x1) ACCUM(p2, v2, x2) ACCUM(p3, v3, x3)
    }
    if (MODE == 1) acc[tid] = dbg;
    __syncthreads();
    // flush: 1024 threads = 1024/RV row lanes
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

template <int D, int MODE>
static int launch_t(const int32_t *pack, const float *val, const int32_t *w_start, const int32_t *w_steps, const float *X,
                    float *Y, const int32_t *fptr, const int32_t *frow, const int32_t *fstart, const int32_t *fn,
                    int n_slots, int n_blocks, int sync_every, void *stream) {
    const size_t lds = (size_t)n_slots * D * 4;
    hipFuncSetAttribute((const void *)spmm_ldsacc_kernel<D, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((spmm_ldsacc_kernel<D, MODE>), dim3(n_blocks), dim3(1024), lds, (hipStream_t)stream, pack, val,
                       w_start, w_steps, X, Y, fptr, frow, fstart, fn, n_slots, sync_every);
    return (int)hipGetLastError();
}

extern "C" int launch_ldsacc(const int32_t *pack, const float *val, const int32_t *w_start, const int32_t *w_steps,
                             const float *X, float *Y, const int32_t *fptr, const int32_t *frow, const int32_t *fstart,
                             const int32_t *fn, int n_slots, int n_blocks, int d, int mode, int sync_every, void *stream) {
#define ARGS pack, val, w_start, w_steps, X, Y, fptr, frow, fstart, fn, n_slots, n_blocks, sync_every, stream
    if (d == 64) return mode == 0 ? launch_t<64, 0>(ARGS) : mode == 1 ? launch_t<64, 1>(ARGS) : launch_t<64, 2>(ARGS);
    if (d == 32) return launch_t<32, 0>(ARGS);
    return 1;
}
