// Fused gather + dot + softplus BPR loss and the gather backward (scatter-add) on gfx950.
//
// Replaces, per training step, the three row gathers (reference
// models/general_cf/lightgcn.py:49-51), cal_bpr_loss (models/loss_utils.py:7-10:
// 2 mul + 2 sum + sub + softplus + sum = 7 launches) and their autograd backward
// (index_put scatter-adds).  One wavefront handles one (anchor,pos,neg) triple at a time:
// lanes stride over d (coalesced 256-B row reads), a DPP/shuffle tree reduces the two dot
// products, per-workgroup partial sums go to a small workspace that the last workgroup to finish
// adds in a fixed order (deterministic loss, no float atomics, one launch).
#include "common.h"
#include "det_scatter.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BPR_BLOCKS 256   // grid-stride workgroups in the forward pass

__device__ __forceinline__ float bpr_term(float x, int variant) {
    if (variant == 0) {
        // F.softplus(beta=1, threshold=20): x if x > 20 else log1p(exp(x))
        return (x > 20.f) ? x : log1pf(expf(x));
    }
    // LightGCL (lightgcl.py:108): -log(sigmoid(pos - neg)) with x = neg - pos
    return -logf(1.f / (1.f + expf(x)));
}

__device__ __forceinline__ float bpr_dterm(float x, int variant) {
    // d/dx of both forms is sigmoid(x) (softplus saturates to 1 above the threshold)
    if (variant == 0 && x > 20.f) return 1.f;
    return 1.f / (1.f + expf(-x));
}

__device__ __forceinline__ int64_t row_of(const int64_t *idx, int b) { return idx ? idx[b] : (int64_t)b; }

// The block that finishes LAST adds the partials -- thread t adds partials t, t + 256, ..., then a 256-wide tree: a fixed order, so the
// value does not depend on which block that is (and equals what the separate finishing kernel of rounds 1-3 produced) -- and writes
// out[0] = mul * total / div (and out[1] = out[0] + *add_in): one launch instead of two.
// "Last" is found with TWO levels of ticket counters: a block takes a ticket of its group of <= 16 blocks, the last block of a group a
// ticket of the top counter.  (One counter for all blocks was measured first: 1024 increments of one address serialize at ~22 ns each
// -- sumsq_kernel went from 6.8 to 29 us, profiles/r04.)  No fence: the partial is published by a RETURNING device-scope atomic
// exchange that the lane waits for before it takes its ticket, so it has reached the coherence point before any later ticket can
// be observed; the reader uses device-scope atomic loads.  (An agent-scope release fence would write back the XCD's whole L2.)
// ws (floats): [0] top ticket, [32 (g + 1)] ticket of group g, [FIN_PART0 + b] partial of block b; every ticket must be 0 at entry and is
// 0 again afterwards (atomicInc wraps).
//  The ordering above is a property of the HARDWARE, not of the HSA memory model: on gfx942 / gfx950 a returning device-scope
// atomic completes at the coherence point and the sc1 loads of the last block bypass the (non-coherent) L2s.  Any other target must
// use a release on the ticket and an acquire in the last block instead; tests/test_gpu_round5.py hammers the one-launch form against
// the two-launch form (thousands of back-to-back launches) so that a toolchain change cannot break it silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "finish_by_last_block relies on gfx942 / gfx950 atomics completing at the coherence point: use release / acquire on this target"
#endif
#define FIN_GROUPS 64
#define FIN_PART0 (32 * (FIN_GROUPS + 1))
// SSLREC_ONE_LAUNCH_REDUCE=0 (A/B switch, read once): the partials are stored plainly and a second one-workgroup launch adds them.
static bool one_launch_reduce() {
    static const bool on = [] { const char *e = getenv("SSLREC_ONE_LAUNCH_REDUCE"); return !(e && e[0] == '0'); }();
    return on;
}

__global__ __launch_bounds__(256) void finish_partials_kernel(const float *ws, int n_blocks, float mul, float div, const float *add_in, float *out) {
    __shared__ float s[256];
    const float *part = ws + FIN_PART0;
    float v = 0.f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) v += part[i];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float r = (mul * s[0]) / div;
        out[0] = r;
        if (add_in) out[1] = r + add_in[0];
    }
}

__device__ __forceinline__ void finish_by_last_block(float *ws, float block_partial, int n_blocks, float mul, float div, const float *add_in,
                                                     float *out, int one_launch) {
    __shared__ float s[256];
    __shared__ int is_last;
    float *part = ws + FIN_PART0;
    if (!one_launch) {
        if (threadIdx.x == 0) part[blockIdx.x] = block_partial;
        return;
    }
    if (threadIdx.x == 0) {
        const int gs = (n_blocks + FIN_GROUPS - 1) / FIN_GROUPS, g = blockIdx.x / gs, n_groups = (n_blocks + gs - 1) / gs;
        const int size_g = min(gs, n_blocks - g * gs);
        (void)__hip_atomic_exchange(part + blockIdx.x, block_partial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (atomicInc(reinterpret_cast<unsigned *>(ws + 32 * (g + 1)), (unsigned)size_g - 1u) == (unsigned)size_g - 1u)
            last = atomicInc(reinterpret_cast<unsigned *>(ws), (unsigned)n_groups - 1u) == (unsigned)n_groups - 1u;
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    float v = 0.f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) v += __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float r = (mul * s[0]) / div;
        out[0] = r;
        if (add_in) out[1] = r + add_in[0];
    }
}

__global__ __launch_bounds__(256) void bpr_fwd_kernel(const float *Ta, const int64_t *ia, const float *Tp,
                                                      const int64_t *ip, const float *Tn, const int64_t *in,
                                                      int B, int d, int variant, float *ws, float divisor, const float *add_in, float *out,
                                                      int one_launch) {
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    float local = 0.f;   // lane 0 of each wave accumulates
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        const float *a = Ta + row_of(ia, b) * d;
        const float *p = Tp + row_of(ip, b) * d;
        const float *n = Tn + row_of(in, b) * d;
        float dp = 0.f, dn = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float av = a[k];
            dp = fmaf(av, p[k], dp);
            dn = fmaf(av, n[k], dn);
        }
        dp = wave_sum(dp);
        dn = wave_sum(dn);
        local += bpr_term(dn - dp, variant);
    }
    if (lane == 0) wsum[w] = local;
    __syncthreads();
    finish_by_last_block(ws, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]), gridDim.x, 1.f, divisor, add_in, out, one_launch);
}

// per sample: coefficient, the three gradient rows (staged in G for indexed roles, stored directly otherwise) and keys
__global__ __launch_bounds__(256) void bpr_bwd_stage_kernel(const float *Ta, const int64_t *ia, const float *Tp,
                                                            const int64_t *ip, const float *Tn, const int64_t *in,
                                                            int B, int d, int variant, float divisor, const float *gscale,
                                                            float *dTa, float *dTp, float *dTn, float *G,
                                                            DetTable tab, int atomic_fallback, f32x4 *zero_tab, size_t zero_n4) {
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    // (kept form) the gradient table the rows will be scattered into is zeroed HERE, under this kernel's latency-bound staging, instead
    // of by a fill launch of its own: nothing in this kernel touches the table (all three roles are staged), the reduction that adds
    // the rows into it is the next launch
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_n4; i += (size_t)gridDim.x * 256) zero_tab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float g = gscale[0] / divisor;
    const int gw = blockIdx.x * 4 + w, nw = gridDim.x * 4;
    for (int b = gw; b < B; b += nw) {
        const int64_t ra = row_of(ia, b), rp = row_of(ip, b), rn = row_of(in, b);
        const float *a = Ta + ra * d;
        const float *p = Tp + rp * d;
        const float *n = Tn + rn * d;
        float dp = 0.f, dn = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float av = a[k];
            dp = fmaf(av, p[k], dp);
            dn = fmaf(av, n[k], dn);
        }
        dp = wave_sum(dp);
        dn = wave_sum(dn);
        const float s = g * bpr_dterm(dn - dp, variant);
        for (int k = lane; k < d; k += 64) {
            const float av = a[k], pv = p[k], nv = n[k];
            const float ga = s * (nv - pv), gp = -s * av, gn = s * av;
            if (atomic_fallback) {
                if (ia) atomicAdd(dTa + ra * d + k, ga); else dTa[ra * d + k] = ga;
                if (ip) atomicAdd(dTp + rp * d + k, gp); else dTp[rp * d + k] = gp;
                if (in) atomicAdd(dTn + rn * d + k, gn); else dTn[rn * d + k] = gn;
            } else {
                if (ia) G[(size_t)(3 * b + 0) * d + k] = ga; else dTa[ra * d + k] = ga;
                if (ip) G[(size_t)(3 * b + 1) * d + k] = gp; else dTp[rp * d + k] = gp;
                if (in) G[(size_t)(3 * b + 2) * d + k] = gn; else dTn[rn * d + k] = gn;
            }
        }
        if (!atomic_fallback && lane < 3) {      // one lane per role; un-indexed roles wrote their row directly
            const int64_t *idx = lane == 0 ? ia : lane == 1 ? ip : in;
            float *dst = lane == 0 ? dTa + ra * d : lane == 1 ? dTp + rp * d : dTn + rn * d;
            if (idx) det_insert(tab, dst, 3 * b + lane); else tab.slot_of[3 * b + lane] = -1;
        }
    }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float *src, const int64_t *idx, int B,
                                                               int d, float *dst) {
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        const int64_t r = idx[b];
        for (int k = lane; k < d; k += 64) atomicAdd(dst + r * d + k, src[(size_t)b * d + k]);
    }
}

__global__ __launch_bounds__(256) void scatter_insert_kernel(const int64_t *idx, int B, int d, float *dst, DetTable tab) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) det_insert(tab, dst + idx[i] * d, i);
}

extern "C" size_t sslrec_bpr_ws_bytes(int32_t B) {
    (void)B;
    return (size_t)(FIN_PART0 + BPR_BLOCKS) * sizeof(float);
}

// backward: staged gradient rows [3B, d] + sort keys
extern "C" size_t sslrec_bpr_bwd_ws_bytes(int32_t B, int32_t d) {
    if (B <= 0 || d <= 0 || 3 * (size_t)B > DET_MAX) return 16;          // atomic fallback: no workspace needed
    return (size_t)3 * B * d * sizeof(float) + det_ws_bytes(3 * B) + 16;
}

extern "C" size_t sslrec_scatter_ws_bytes(int32_t B) {
    if (B <= 0 || B > DET_MAX) return 16;
    return det_ws_bytes(B) + 16;
}

static int bpr_fwd_any(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip, const float *Tn, const int64_t *in, int32_t B,
                       int32_t d, int32_t variant, float divisor, const float *add_in, float *ws, float *loss_out, void *stream) {
    if (!Ta || !Tp || !Tn || !ws || !loss_out || B < 0 || d <= 0 || (variant != 0 && variant != 1) || !(divisor != 0.f))
        return SSLREC_E_BADARG;
    const int one = one_launch_reduce() ? 1 : 0;
    hipLaunchKernelGGL(bpr_fwd_kernel, dim3(BPR_BLOCKS), dim3(256), 0, (hipStream_t)stream, Ta, ia, Tp, ip, Tn, in, B, d,
                       variant, ws, divisor, add_in, loss_out, one);
    SSLREC_LAUNCH_CHECK();
    if (!one) {
        hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, BPR_BLOCKS, 1.f, divisor, add_in, loss_out);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sslrec_bpr_fwd_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                                  const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant,
                                  float divisor, float *ws, float *loss_out, void *stream) {
    return bpr_fwd_any(Ta, ia, Tp, ip, Tn, in, B, d, variant, divisor, nullptr, ws, loss_out, stream);
}

extern "C" int sslrec_bpr_fwd_total_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                                        const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant,
                                        float divisor, const float *add_in, float *ws, float *loss_out2, void *stream) {
    if (!add_in) return SSLREC_E_BADARG;
    return bpr_fwd_any(Ta, ia, Tp, ip, Tn, in, B, d, variant, divisor, add_in, ws, loss_out2, stream);
}

// `kept_clean`: the scatter table inside ws was initialised once (sslrec_bpr_bwd_table_init) and every call since left it clean:
// no clearing launch, the reduction returns every slot it used (det_reduce_kernel, self_clean)
static int bpr_bwd_any(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                       const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant, float divisor,
                       const float *gscale_dev, float *dTa, float *dTp, float *dTn, void *ws, void *stream, bool kept_clean,
                       float *zero_table = nullptr, size_t zero_elems = 0) {
    if (!Ta || !Tp || !Tn || !gscale_dev || !dTa || !dTp || !dTn || B < 0 || d <= 0 ||
        (variant != 0 && variant != 1) || !(divisor != 0.f))
        return SSLREC_E_BADARG;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int blocks = (B + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const bool indexed = ia || ip || in;
    const bool det = indexed && ws && 3 * (size_t)B <= DET_MAX && d <= 256;
    if (!det) {      // nothing to scatter (dense rows), or a batch beyond the table: plain stores / atomic adds
        if (zero_table) return SSLREC_E_BADARG;      // (the fused zero fill needs every role staged: see sslrec_bpr_bwd_kept_f32)
        hipLaunchKernelGGL(bpr_bwd_stage_kernel, dim3(blocks), dim3(256), 0, st, Ta, ia, Tp, ip, Tn, in, B, d, variant, divisor,
                           gscale_dev, dTa, dTp, dTn, (float *)nullptr, DetTable{}, 1, (f32x4 *)nullptr, (size_t)0);
        SSLREC_LAUNCH_CHECK();
        return 0;
    }
    if (zero_table && (!ia || !ip || !in || (zero_elems & 3) || ((uintptr_t)zero_table & 15))) return SSLREC_E_BADARG;
    float *G = (float *)ws;
    const DetTable tab = det_table(G + (size_t)3 * B * d);
    if (!kept_clean) {
        hipLaunchKernelGGL(det_clear_kernel, dim3(64), dim3(256), 0, st, tab);
        SSLREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bpr_bwd_stage_kernel, dim3(blocks), dim3(256), 0, st, Ta, ia, Tp, ip, Tn, in, B, d, variant, divisor,
                       gscale_dev, dTa, dTp, dTn, G, tab, 0, reinterpret_cast<f32x4 *>(zero_table), zero_elems / 4);
    SSLREC_LAUNCH_CHECK();
    return det_reduce(tab, 3 * B, G, d, st, kept_clean ? 1 : 0);
}

extern "C" int sslrec_bpr_bwd_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                                  const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant, float divisor,
                                  const float *gscale_dev, float *dTa, float *dTp, float *dTn, void *ws, void *stream) {
    return bpr_bwd_any(Ta, ia, Tp, ip, Tn, in, B, d, variant, divisor, gscale_dev, dTa, dTp, dTn, ws, stream, false);
}

extern "C" int sslrec_bpr_bwd_table_init(void *ws, int32_t B, int32_t d, void *stream) {
    if (!ws || B <= 0 || d <= 0 || 3 * (size_t)B > DET_MAX || d > 256) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(det_clear_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, det_table((float *)ws + (size_t)3 * B * d));
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_bpr_bwd_kept_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                                       const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant, float divisor,
                                       const float *gscale_dev, float *dTa, float *dTp, float *dTn, void *ws, float *zero_table,
                                       size_t zero_elems, void *stream) {
    if (!ws || 3 * (size_t)B > DET_MAX || d > 256) return SSLREC_E_BADARG;      // (this form exists for the table it keeps clean)
    return bpr_bwd_any(Ta, ia, Tp, ip, Tn, in, B, d, variant, divisor, gscale_dev, dTa, dTp, dTn, ws, stream, true, zero_table, zero_elems);
}

extern "C" int sslrec_scatter_add_rows_f32(const float *src, const int64_t *idx, int32_t B, int32_t d,
                                           float *dst, void *ws, void *stream) {
    if (!src || !idx || !dst || B < 0 || d <= 0) return SSLREC_E_BADARG;
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!ws || B > DET_MAX || d > 256) {
        int blocks = (B + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks), dim3(256), 0, st, src, idx, B, d, dst);
        SSLREC_LAUNCH_CHECK();
        return 0;
    }
    const DetTable tab = det_table(ws);
    hipLaunchKernelGGL(det_clear_kernel, dim3(64), dim3(256), 0, st, tab);
    SSLREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter_insert_kernel, dim3((B + 255) / 256), dim3(256), 0, st, idx, B, d, dst, tab);
    SSLREC_LAUNCH_CHECK();
    return det_reduce(tab, B, src, d, st);
}

// ---- L2 regularizer term: sum of squares of a parameter table (reference models/loss_utils.py:20-24,
// `W.norm(2).square()` = norm + square per parameter, plus their autograd) ---------------------------
#define SUMSQ_BLOCKS 1024

__global__ __launch_bounds__(256) void sumsq_kernel(const float *x, size_t n, float *ws, float weight, float *out, int one_launch) {
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    const size_t n4 = n / 4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        acc = fmaf(v[0], v[0], acc);
        acc = fmaf(v[1], v[1], acc);
        acc = fmaf(v[2], v[2], acc);
        acc = fmaf(v[3], v[3], acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail (n not a multiple of 4)
        const float v = x[n4 * 4 + threadIdx.x];
        acc = fmaf(v, v, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    finish_by_last_block(ws, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]), gridDim.x, weight, 1.f, nullptr, out, one_launch);
}

// out = (2 * g * weight) * x : gradient of g * weight * sum(x^2)
__global__ __launch_bounds__(256) void scale2_kernel(const float *x, size_t n, const float *g, float weight, float *out) {
    const float s = 2.f * (g[0] * weight);
    const size_t n4 = n / 4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    f32x4 *o4 = reinterpret_cast<f32x4 *>(out);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = x4[i];
        v *= s;
        o4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = s * x[n4 * 4 + threadIdx.x];
}

extern "C" size_t sslrec_sumsq_ws_bytes(void) { return (size_t)(FIN_PART0 + SUMSQ_BLOCKS) * sizeof(float); }

extern "C" int sslrec_sumsq_fwd_f32(const float *x, size_t n, float weight, float *ws, float *out, void *stream) {
    if (!x || !ws || !out || ((uintptr_t)x & 15)) return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int one = one_launch_reduce() ? 1 : 0;
    hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, st, x, n, ws, weight, out, one);
    SSLREC_LAUNCH_CHECK();
    if (!one) {
        hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(256), 0, st, ws, SUMSQ_BLOCKS, weight, 1.f, (const float *)nullptr, out);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sslrec_sumsq_bwd_f32(const float *x, size_t n, float weight, const float *gscale_dev, float *dx, void *stream) {
    if (!x || !gscale_dev || !dx || ((uintptr_t)x & 15) || ((uintptr_t)dx & 15)) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(scale2_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, x, n, gscale_dev, weight, dx);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// Adam step over a parameter table (replaces torch.optim.Adam as the reference's Trainer uses it,
// trainer/trainer.py:45-49,68: lr, weight_decay as L2 added to the gradient, betas (0.9, 0.999), eps 1e-8).
// One pass: 16 B read + 12 B written per element instead of ~10 elementwise launches.
// state (device, 4 floats): [0] step count (as float bits of an int), [1] lr / (1 - b1^t), [2] sqrt(1 - b2^t)
// -- the tick kernel advances t and computes the two scalars in double like the reference's Python does, so a
// captured hipGraph replays the right bias correction without host involvement.
// ---------------------------------------------------------------------------------------
__global__ void adam_tick_kernel(float *state, double lr, double b1, double b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int t = __float_as_int(state[0]) + 1;
    state[0] = __int_as_float(t);
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    state[1] = (float)(lr / bc1);
    state[2] = (float)sqrt(bc2);
}

__global__ __launch_bounds__(256) void adam_apply_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                         float *__restrict__ m, float *__restrict__ v, size_t n,
                                                         const float *__restrict__ state, float w1, float b2,
                                                         float w2, float eps, float wd) {
    const float step_size = state[1], bc2_sqrt = state[2];      // w1 = 1-beta1, w2 = 1-beta2, rounded from double
    const size_t n4 = n / 4;
    f32x4 *p4 = reinterpret_cast<f32x4 *>(p);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    f32x4 *m4 = reinterpret_cast<f32x4 *>(m);
    f32x4 *v4 = reinterpret_cast<f32x4 *>(v);
#define ADAM_ONE(PP, GG, MM, VV)                                   \
    {                                                              \
        const float gg = (wd != 0.f) ? fmaf(wd, PP, GG) : GG;      \
        MM = MM + w1 * (gg - MM);                                  \
        VV = VV * b2 + w2 * (gg * gg);                             \
        const float denom = sqrtf(VV) / bc2_sqrt + eps;            \
        PP = PP - step_size * (MM / denom);                        \
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
        const f32x4 gv = g4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) ADAM_ONE(pp[k], gv[k], mm[k], vv[k])
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float pp = p[i], mm = m[i], vv = v[i];
        ADAM_ONE(pp, g[i], mm, vv)
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
#undef ADAM_ONE
}

extern "C" int sslrec_adam_tick(float *state, double lr, double beta1, double beta2, void *stream) {
    if (!state || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_adam_apply_f32(float *p, const float *g, float *m, float *v, size_t n, const float *state,
                                     double beta1, double beta2, double eps, double weight_decay, void *stream) {
    if (!p || !g || !m || !v || !state) return SSLREC_E_BADARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return SSLREC_E_BADARG;
    if (n == 0) return 0;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, state,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------
// Rank-q products of LightGCL's SVD view (replaces `u_mul_s @ (vt @ E)` and its autograd, models/general_cf/
// lightgcl.py:83-84: q = 5 singular triplets): a tall-skinny reduction S[q,d] = M^T X over N rows and the matching
// expansion Y[N,d] = M S.  rocBLAS picks a 64x16x256 macro-tile for these shapes and needs 125-195 us per call at
// N = 92 k, d = 64; both are one streaming pass over an N x d table (37 MB).  M(qi, n) = M[qi*sq + n*sn] covers the
// row-major [q,N] factor (sq = N, sn = 1) and the row-major [N,q] factor (sq = 1, sn = q).
// ---------------------------------------------------------------------------------------
#define RANKQ_MAX 16
#define RANKQ_BLOCKS 256

// Fast path (d / 4 divides 64): the table is a flat stream of float4; a lane's column group never changes along its
// grid-stride walk (the stride is a multiple of d / 4), so it keeps q float4 accumulators, RANKQ_UNROLL loads in flight.
// Lanes of a wave with the same column group are folded by shuffles, the block's 4 waves through LDS: one partial
// [q, d] per block, summed in fixed order by rankq_sum_kernel (deterministic).
#define RANKQ_UNROLL 4
template <int Q>
__global__ __launch_bounds__(256) void rankq_reduce_vec_kernel(const float *__restrict__ M, long sq, long sn,
                                                               const float4 *__restrict__ X, long n_elem, int dv,
                                                               float *__restrict__ partial) {
    __shared__ float4 red[4][Q][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long stride = (long)gridDim.x * 256;
    float4 acc[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long e0 = (long)blockIdx.x * 256 + tid; e0 < n_elem; e0 += stride * RANKQ_UNROLL) {
        float4 x[RANKQ_UNROLL];
        float m[RANKQ_UNROLL][Q];
#pragma unroll
        for (int u = 0; u < RANKQ_UNROLL; ++u) {
            const long e = e0 + u * stride;
            const bool ok = e < n_elem;
            x[u] = ok ? X[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            const long n = ok ? e / dv : 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) m[u][k] = M[k * sq + n * sn];
        }
#pragma unroll
        for (int u = 0; u < RANKQ_UNROLL; ++u)
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                acc[k].x = fmaf(m[u][k], x[u].x, acc[k].x); acc[k].y = fmaf(m[u][k], x[u].y, acc[k].y);
                acc[k].z = fmaf(m[u][k], x[u].z, acc[k].z); acc[k].w = fmaf(m[u][k], x[u].w, acc[k].w);
            }
    }
    // lanes lane, lane + dv, lane + 2 dv, ... hold the same column group
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        for (int o = 32; o >= dv; o >>= 1) {
            acc[k].x += __shfl_xor(acc[k].x, o, 64); acc[k].y += __shfl_xor(acc[k].y, o, 64);
            acc[k].z += __shfl_xor(acc[k].z, o, 64); acc[k].w += __shfl_xor(acc[k].w, o, 64);
        }
        red[w][k][lane] = acc[k];
    }
    __syncthreads();
    for (int i = tid; i < Q * dv; i += 256) {
        const int k = i / dv, c = i % dv;
        float4 t = red[0][k][c];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float4 v = red[ww][k][c];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        reinterpret_cast<float4 *>(partial)[((size_t)blockIdx.x * Q + k) * dv + c] = t;
    }
}

template <int Q>
__global__ __launch_bounds__(256) void rankq_expand_vec_kernel(const float *__restrict__ M, long sq, long sn,
                                                               const float4 *__restrict__ S, long n_elem, int dv,
                                                               float4 *__restrict__ Y) {
    const int tid = threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const long e_first = (long)blockIdx.x * 256 + tid;
    float4 s[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) s[k] = S[k * dv + (int)(e_first % dv)];
    for (long e0 = e_first; e0 < n_elem; e0 += stride * RANKQ_UNROLL) {
        float m[RANKQ_UNROLL][Q];
#pragma unroll
        for (int u = 0; u < RANKQ_UNROLL; ++u) {
            const long e = e0 + u * stride;
            const long n = e < n_elem ? e / dv : 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) m[u][k] = M[k * sq + n * sn];
        }
#pragma unroll
        for (int u = 0; u < RANKQ_UNROLL; ++u) {
            const long e = e0 + u * stride;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                y.x = fmaf(m[u][k], s[k].x, y.x); y.y = fmaf(m[u][k], s[k].y, y.y);
                y.z = fmaf(m[u][k], s[k].z, y.z); y.w = fmaf(m[u][k], s[k].w, y.w);
            }
            if (e < n_elem) Y[e] = y;
        }
    }
}

__global__ __launch_bounds__(256) void rankq_reduce_kernel(const float *__restrict__ M, long sq, long sn,
                                                           const float *__restrict__ X, int N, int d, int q,
                                                           float *__restrict__ partial) {
    // one wave per slice of rows; lane j owns columns j, j+64, ... of its rows; q accumulators per owned column
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block();
    const int n_waves = gridDim.x * 4;
    const int per = (N + n_waves - 1) / n_waves;
    const int n0 = w * per, n1 = min(N, n0 + per);
    for (int c0 = 0; c0 < d; c0 += 64) {
        const int c = c0 + lane;
        float acc[RANKQ_MAX];
#pragma unroll
        for (int k = 0; k < RANKQ_MAX; ++k) acc[k] = 0.f;
        if (c < d) {
            for (int n = n0; n < n1; ++n) {
                const float x = X[(size_t)n * d + c];
#pragma unroll
                for (int k = 0; k < RANKQ_MAX; ++k)
                    if (k < q) acc[k] = fmaf(M[k * sq + n * sn], x, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < RANKQ_MAX; ++k)
                if (k < q) partial[((size_t)w * q + k) * d + c] = acc[k];
        }
    }
}

// out[i] = sum over the partials of all waves: one wave per output, lanes stride over the partials, butterfly at the end
// (fixed order: deterministic)
__global__ __launch_bounds__(256) void rankq_sum_kernel(const float *__restrict__ partial, int n_parts, int qd,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave_in_block();
    if (i >= qd) return;
    float s = 0.f;
    for (int p = lane; p < n_parts; p += 64) s += partial[(size_t)p * qd + i];
    s = wave_sum(s);
    if (lane == 0) out[i] = s;
}

__global__ __launch_bounds__(256) void rankq_expand_kernel(const float *__restrict__ M, long sq, long sn,
                                                           const float *__restrict__ S, int N, int d, int q,
                                                           float *__restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block();
    const int n_waves = gridDim.x * 4;
    for (int c0 = 0; c0 < d; c0 += 64) {
        const int c = c0 + lane;
        if (c >= d) continue;
        float s[RANKQ_MAX];
#pragma unroll
        for (int k = 0; k < RANKQ_MAX; ++k) s[k] = (k < q) ? S[k * d + c] : 0.f;
        for (int n = w; n < N; n += n_waves) {
            float y = 0.f;
#pragma unroll
            for (int k = 0; k < RANKQ_MAX; ++k)
                if (k < q) y = fmaf(M[k * sq + n * sn], s[k], y);
            Y[(size_t)n * d + c] = y;
        }
    }
}

// the vector kernels: d / 4 a power of two <= 64 (d = 4 ... 256) and q small enough for register-resident factors
static bool rankq_vec_ok(int d, int q) {
    const int dv = d / 4;
    return d % 4 == 0 && dv >= 1 && dv <= 64 && (dv & (dv - 1)) == 0 && q <= 8;
}
#define RANKQ_DISPATCH(q, GO) \
    switch (q) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break; \
                 case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); break; }

extern "C" size_t sslrec_rankq_ws_bytes(int32_t q, int32_t d) {
    if (q <= 0 || d <= 0) return 0;
    return (size_t)RANKQ_BLOCKS * 4 * q * d * sizeof(float);
}

extern "C" int sslrec_rankq_reduce_f32(const float *M, int64_t stride_q, int64_t stride_n, const float *X, int32_t N,
                                       int32_t d, int32_t q, float *ws, float *out, void *stream) {
    if (!M || !X || !ws || !out || N <= 0 || d <= 0 || q <= 0 || q > RANKQ_MAX) return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int dv = d / 4;
    int n_parts = RANKQ_BLOCKS * 4;
    if (rankq_vec_ok(d, q) && (((uintptr_t)X | (uintptr_t)ws) & 15) == 0) {      // one partial per block of the vector kernel
        const long n_elem = (long)N * dv;
        n_parts = (int)((n_elem + 1023) / 1024 < RANKQ_BLOCKS * 4 ? (n_elem + 1023) / 1024 : RANKQ_BLOCKS * 4);
#define RQ_GO(QQ) hipLaunchKernelGGL(rankq_reduce_vec_kernel<QQ>, dim3(n_parts), dim3(256), 0, st, M, (long)stride_q, (long)stride_n, \
                                     reinterpret_cast<const float4 *>(X), n_elem, dv, ws)
        RANKQ_DISPATCH(q, RQ_GO)
#undef RQ_GO
    } else {
        hipLaunchKernelGGL(rankq_reduce_kernel, dim3(RANKQ_BLOCKS), dim3(256), 0, st, M, (long)stride_q, (long)stride_n, X, N, d, q,
                           ws);
    }
    SSLREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(rankq_sum_kernel, dim3((q * d + 3) / 4), dim3(256), 0, st, ws, n_parts, q * d, out);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_rankq_expand_f32(const float *M, int64_t stride_q, int64_t stride_n, const float *S, int32_t N,
                                       int32_t d, int32_t q, float *Y, void *stream) {
    if (!M || !S || !Y || N <= 0 || d <= 0 || q <= 0 || q > RANKQ_MAX) return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (rankq_vec_ok(d, q) && ((uintptr_t)S & 15) == 0 && ((uintptr_t)Y & 15) == 0) {
        const int dv = d / 4;
        const long n_elem = (long)N * dv;
        const int blocks = (int)((n_elem + 1023) / 1024 < RANKQ_BLOCKS * 8 ? (n_elem + 1023) / 1024 : RANKQ_BLOCKS * 8);
#define RQ_GO(QQ) hipLaunchKernelGGL(rankq_expand_vec_kernel<QQ>, dim3(blocks), dim3(256), 0, st, M, (long)stride_q, (long)stride_n, \
                                     reinterpret_cast<const float4 *>(S), n_elem, dv, reinterpret_cast<float4 *>(Y))
        RANKQ_DISPATCH(q, RQ_GO)
#undef RQ_GO
    } else {
        hipLaunchKernelGGL(rankq_expand_kernel, dim3(RANKQ_BLOCKS), dim3(256), 0, st, M, (long)stride_q, (long)stride_n, S, N, d, q, Y);
    }
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---- the rows a fused BPR backward writes, as a bitmap (the zero-row hint of the first backward product, sslrec_epilogue_t) ----------
__global__ __launch_bounds__(1024) void row_bits3_kernel(const int64_t *__restrict__ i0, long o0, const int64_t *__restrict__ i1, long o1,
                                                         const int64_t *__restrict__ i2, long o2, int n, int n_words, uint32_t *__restrict__ bits) {
    extern __shared__ uint32_t rb_lds[];
    for (int w = threadIdx.x; w < n_words; w += 1024) rb_lds[w] = 0u;
    __syncthreads();
    const long n_bits = (long)n_words * 32;              // (an index outside the table sets nothing: the caller's gather would have faulted on it)
    for (int i = threadIdx.x; i < n; i += 1024) {
        const long r0 = i0[i] + o0;
        if (r0 >= 0 && r0 < n_bits) atomicOr(&rb_lds[r0 >> 5], 1u << (r0 & 31));
        if (i1) { const long r = i1[i] + o1; if (r >= 0 && r < n_bits) atomicOr(&rb_lds[r >> 5], 1u << (r & 31)); }
        if (i2) { const long r = i2[i] + o2; if (r >= 0 && r < n_bits) atomicOr(&rb_lds[r >> 5], 1u << (r & 31)); }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < n_words; w += 1024) bits[w] = rb_lds[w];
}

extern "C" int sslrec_row_bits3(const int64_t *idx0, int64_t off0, const int64_t *idx1, int64_t off1, const int64_t *idx2, int64_t off2,
                                int32_t n, int32_t n_rows, uint32_t *bits, void *stream) {
    if (!idx0 || n < 0 || n_rows <= 0 || n_rows > (1 << 20) || !bits) return SSLREC_E_BADARG;
    const int n_words = (n_rows + 31) / 32;
    const size_t lds = (size_t)n_words * sizeof(uint32_t);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)row_bits3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(row_bits3_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, idx0, (long)off0, idx1, (long)off1, idx2, (long)off2,
                       (int)n, n_words, bits);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---- step-level scalar / table helpers of the one-node contrastive steps (ops.contrastive_step; include/sslrec_hip.h) ----------------
__global__ void weighted_sum4_kernel(const float *t0, float w0, const float *t1, float w1, const float *t2, float w2, const float *t3, float w3,
                                     float *out) {
    const float a = t0 ? w0 * t0[0] : 0.f, b = t1 ? w1 * t1[0] : 0.f, c = t2 ? w2 * t2[0] : 0.f, e = t3 ? w3 * t3[0] : 0.f;
    out[1] = a; out[2] = b; out[3] = c; out[4] = e;
    out[0] = ((a + b) + c) + e;
    out[5] = b + c;
}

extern "C" int sslrec_weighted_sum4_f32(const float *t0, float w0, const float *t1, float w1, const float *t2, float w2, const float *t3,
                                        float w3, float *out6, void *stream) {
    if (!out6) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(weighted_sum4_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t0, w0, t1, w1, t2, w2, t3, w3, out6);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

__global__ void scalar_scale2_kernel(const float *x, float a, float b, float *out) { out[0] = a * x[0]; out[1] = b * x[0]; }

extern "C" int sslrec_scalar_scale2_f32(const float *x, float a, float b, float *out2, void *stream) {
    if (!x || !out2) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(scalar_scale2_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, x, a, b, out2);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void add_tables_kernel(const float4 *a, const float4 *b, const float4 *c, float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = a[i];
        const float4 y = b[i];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        if (c) { const float4 z = c[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
        out[i] = x;
    }
}

extern "C" int sslrec_add_tables_f32(const float *a, const float *b, const float *c, float *out, size_t n, void *stream) {
    if (!a || !b || !out || (n & 3) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15)) return SSLREC_E_BADARG;
    if (n == 0) return 0;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    hipLaunchKernelGGL(add_tables_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4 *>(a),
                       reinterpret_cast<const float4 *>(b), reinterpret_cast<const float4 *>(c), reinterpret_cast<float4 *>(out), n4);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_abi_version(void) { return SSLREC_ABI_VERSION; }
