// Fused InfoNCE (contrastive loss against ALL rows of a view) for MI355X (gfx950).
//
// Replaces cal_infonce_loss (reference models/loss_utils.py:30-39; call sites
// models/general_cf/simgcl.py:49, sgl.py:57-59) and LightGCL's un-normalized variant
// (models/general_cf/lightgcl.py:114-118).  The reference materializes the B x M score
// matrix three times in the forward pass (mm, /temp, exp) and keeps it for autograd
// (4096 x 91,599 fp32 = 1.5 GB each); here it never leaves the register file:
//
//   forward   rowsum[b] = sum_j exp2( <e1s_b, a_j> )          e1s = e1^ * log2(e)/temp
//   backward  W[b,:]    = sum_j exp2(.) a_j                   (grad wrt anchors)
//             dA[j,:]   = sum_b exp2(.) V[b,:]                (grad wrt all rows)
//
// The products run on the matrix cores in one of two arithmetics (SSLREC_INFONCE_PRECISION, see inf_precision):
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak; the kernels in this file), or bf16 MFMA over THREE bf16
// planes per operand with the 6 significant cross terms (infonce_x3.inc: fp32-equivalent error at 6/16 of the cost,
// the default) -- plain bf16 would break the fp32 parity the north star asks for.  One wavefront per SIMD
// (4 per CU) keeps its 128 anchors' operand (or its 64-128 "all" rows) resident in VGPRs
// and streams the other operand straight from L2 in the MFMA fragment layout, so there is
// no LDS staging and no barrier in the hot loops:
//   * operand fragment of a 32-row tile: lane (r = lane&31, h = lane>>5) holds row r,
//     elements [h*d/2, (h+1)*d/2)  -> 128 contiguous bytes per lane at d=64 (float4 loads);
//     MFMA k-step kk contracts elements {kk, d/2+kk}: both operands use the same
//     permutation of k, so the dot product is unchanged.
//   * scores are computed TRANSPOSED where the row sum is wanted (rows = all-index j, cols =
//     anchors): an anchor's 16 partial scores then sit in one lane, exp2 + add needs no
//     cross-lane traffic, and the C-layout registers feed the second MFMA directly as its
//     B operand (k-step r <-> C register r).
// The un-subtracted exp matches the reference (no max-subtraction there either); with
// normalized rows |score| <= 1/temp.
#include "common.h"
#include "det_scatter.h"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

template <int D> struct IC {
    static constexpr int HALF = D / 2;                  // MFMA k-steps per 32x32 tile (K=2 each)
    static constexpr int NDT = D / 32;                  // 32-wide tiles along d
    static constexpr int TA = (D == 128) ? 2 : 4;       // 32-row tiles a wave keeps resident
    static constexpr int ROWS = TA * 32;                // resident rows per wave
};

// row inside a 32x32 C tile that register r of a lane with half-index h holds
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// k-major operand fragment of rows [row0, row0+32): rows >= limit are clamped to limit-1
template <int D>
__device__ __forceinline__ void load_frag(float (&f)[D / 2], const float *__restrict__ base, int row0, int limit,
                                          int lane) {
    constexpr int HALF = D / 2;
    int row = row0 + (lane & 31);
    row = (row < limit) ? row : (limit - 1);
    const f32x4 *p = reinterpret_cast<const f32x4 *>(base + (size_t)row * D + (lane >> 5) * HALF);
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
        const f32x4 v = p[q];
        f[4 * q + 0] = v[0];
        f[4 * q + 1] = v[1];
        f[4 * q + 2] = v[2];
        f[4 * q + 3] = v[3];
    }
}

// transposed fragment for the second product: step r of lane (c, h) holds
// base[row0 + crow(r,h)][dt*32 + c]; rows >= limit read as 0
template <int D>
__device__ __forceinline__ void load_frag_t(float (&f)[16], const float *__restrict__ base, int row0, int limit,
                                            int dt, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + crow(r, h);
        f[r] = (row < limit) ? base[(size_t)row * D + dt * 32 + c] : 0.f;
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

template <int HALF>
__device__ __forceinline__ f32x16 tile_dot(const float (&a)[HALF], const float (&b)[HALF]) {
    f32x16 acc = zero16();
#pragma unroll
    for (int kk = 0; kk < HALF; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------
// row preparation: dst[r,:] = scale * x / sqrt(1e-8 + |x|^2)  (or scale * x), rn[r] = 1/sqrt(..)
// one wavefront per row; x = src[idx ? idx[r] : r]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_rows_kernel(const float *src, const int64_t *idx, int n, int d,
                                                        int do_norm, float scale, float *dst, float *rn) {
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    for (int r = blockIdx.x * 4 + w; r < n; r += gridDim.x * 4) {
        const float *x = src + (idx ? idx[r] : (int64_t)r) * d;
        float inv = 1.f;
        if (do_norm) {
            float ss = 0.f;
            for (int k = lane; k < d; k += 64) ss = fmaf(x[k], x[k], ss);
            ss = wave_sum(ss);
            inv = 1.f / sqrtf(1e-8f + ss);
        }
        for (int k = lane; k < d; k += 64) dst[(size_t)r * d + k] = (x[k] * inv) * scale;
        if (lane == 0 && rn) rn[r] = inv;
    }
}

// ---------------------------------------------------------------------------------------
// forward hot kernel: zpart[split][b] = sum_{j in split} exp2(<e1s_b, a_j>)
// block = 4 waves = 4 x ROWS anchors sharing one column split (same a_j stream -> L1/L2 hits)
// ---------------------------------------------------------------------------------------
// TA = anchor tiles resident per wave.  TA=2 keeps the kernel under 128 registers per lane so that TWO
// waves share a SIMD: while one wave's dependent MFMA chain paces the matrix pipe, the other issues its
// exp2 / row-sum epilogue and its operand loads (the one-wave version left those un-overlapped).
template <int D, int TA>
__global__ __launch_bounds__(256, (TA <= 2 && D <= 64) ? 2 : 1) void infonce_rowsum_kernel(const float *__restrict__ E1s,
                                                                const float *__restrict__ An, int B, int M,
                                                                int n_agroup, int cols_per_split,
                                                                float *__restrict__ zpart) {
    constexpr int HALF = IC<D>::HALF, ROWS = TA * 32;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int ag = blockIdx.x % n_agroup;
    const int split = blockIdx.x / n_agroup;
    const int a0 = (ag * 4 + wave_in_block()) * ROWS;
    if (a0 >= B) return;
    const int j_begin = split * cols_per_split;
    int j_end = j_begin + cols_per_split;
    if (j_end > M) j_end = M;

    float e1[TA][HALF];
#pragma unroll
    for (int t = 0; t < TA; ++t) load_frag<D>(e1[t], E1s, a0 + t * 32, B, lane);
    float rs[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) rs[t] = 0.f;

    float an[HALF];
    if (j_begin < j_end) load_frag<D>(an, An, j_begin, M, lane);
    for (int j0 = j_begin; j0 < j_end; j0 += 32) {
        float nx[HALF];
        const bool more = (j0 + 32 < j_end);
        if (more) load_frag<D>(nx, An, j0 + 32, M, lane);   // prefetch next tile of "all" rows
        const bool full = (j0 + 32 <= j_end);
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const f32x16 s = tile_dot<HALF>(an, e1[t]);        // s[j][anchor], transposed scores
            float part = 0.f;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) part += __builtin_amdgcn_exp2f(s[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    part += (j0 + crow(r, h) < j_end) ? __builtin_amdgcn_exp2f(s[r]) : 0.f;
            }
            rs[t] += part;
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < HALF; ++k) an[k] = nx[k];
        }
    }
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const float tot = rs[t] + __shfl_xor(rs[t], 32, 64);
        const int b = a0 + t * 32 + (lane & 31);
        if (h == 0 && b < B) zpart[(size_t)split * B + b] = tot;
    }
}

// ---------------------------------------------------------------------------------------
// forward finish: Z_b = sum_split zpart, loss_b = -pos_b + log(Z_b [+1e-8]); block partials
// ---------------------------------------------------------------------------------------
// fin (round 6; null = the two-launch form, partials[] then infonce_reduce_kernel): the workgroup that finishes LAST adds the per-workgroup
// partial losses in a fixed order, takes min_b Z_b and writes out[0] / misc[0] -- the reduce launch is gone.  Same two-level ticket
// scheme and the same hardware argument as finish_by_last_block of losses.hip (returning device-scope atomics complete at the coherence
// point on gfx942 / gfx950; the last block reads with device-scope atomic loads).  fin (floats): [0] top ticket, [32 (g + 1)] ticket of
// group g < FINF_GROUPS, [FINF_PART0 + b] partial loss of block b, [FINF_ZMIN0 + b] its min Z.  Tickets are zeroed by the call's
// preparation launch and are 0 again afterwards (atomicInc wraps).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the one-launch InfoNCE finish relies on gfx942 / gfx950 atomics completing at the coherence point"
#endif
#define FINF_GROUPS 16
#define FINF_PART0 (32 * (FINF_GROUPS + 1))
#define FINF_ZMIN0 (FINF_PART0 + 256)
#define FINF_FLOATS (FINF_ZMIN0 + 256)
__global__ __launch_bounds__(256) void infonce_finish_fwd_kernel(const float *E1s, const float *E2n,
                                                                 const float *zpart, int n_split, int B, int d,
                                                                 int variant, float *Z, float *partials, float *fin, float *out, float *misc,
                                                                 const float *dyn_bias) {
    __shared__ float wsum[4], wmin[4];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    float local = 0.f, zmin = 3.0e38f;
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        float z = 0.f;
        for (int s = lane; s < n_split; s += 64) z += zpart[(size_t)s * B + b];
        z = wave_sum(z);
        float dot = 0.f;
        for (int k = lane; k < d; k += 64) dot = fmaf(E1s[(size_t)b * d + k], E2n[(size_t)b * d + k], dot);
        float pos = wave_sum(dot) * LN2_F;             // = <e1^, e2^> / temp
        float lz;
        if (variant == 0) {
            lz = logf(z);
        } else {
            pos = fminf(fmaxf(pos, -5.f), 5.f);
            lz = logf(z + 1e-8f);
        }
        if (lane == 0) Z[b] = z;
        // (dyn: the backward's V'_b = 2^-bias_b V_b is bounded through min_b (Z_b + eps) 2^bias_b)
        zmin = fminf(zmin, dyn_bias ? (z + 1e-8f) * exp2f(dyn_bias[b]) : z);
        local += lz - pos;
    }
    if (lane == 0) { wsum[w] = local; wmin[w] = zmin; }
    __syncthreads();
    if (!fin) {
        if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        return;
    }
    const int n_blocks = gridDim.x;
    if (threadIdx.x == 0) {
        const float bp = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        const float bm = fminf(fminf(wmin[0], wmin[1]), fminf(wmin[2], wmin[3]));
        const int gs = (n_blocks + FINF_GROUPS - 1) / FINF_GROUPS, g = blockIdx.x / gs, n_groups = (n_blocks + gs - 1) / gs;
        const int size_g = min(gs, n_blocks - g * gs);
        (void)__hip_atomic_exchange(fin + FINF_PART0 + blockIdx.x, bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_exchange(fin + FINF_ZMIN0 + blockIdx.x, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (atomicInc(reinterpret_cast<unsigned *>(fin + 32 * (g + 1)), (unsigned)size_g - 1u) == (unsigned)size_g - 1u)
            last = atomicInc(reinterpret_cast<unsigned *>(fin), (unsigned)n_groups - 1u) == (unsigned)n_groups - 1u;
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    // the sums of infonce_reduce_kernel, in its order (thread t adds partials t, t + 256, ...; n_blocks <= 256 here: one each)
    __shared__ float s[256];
    float v = 0.f, m = 3.0e38f;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        v += __hip_atomic_load(fin + FINF_PART0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m = fminf(m, __hip_atomic_load(fin + FINF_ZMIN0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
    __syncthreads();
    s[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] = fminf(s[threadIdx.x], s[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && misc) misc[0] = s[0];
}

// (also leaves min_b Z_b in misc[0]: the h3 mode's backward chooses the fp16 scale of V = g ln2 / Z_b * e1s_b from it)
__global__ __launch_bounds__(256) void infonce_reduce_kernel(const float *partials, int n, float *out, const float *Z, int B, float *misc) {
    __shared__ float s[256];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
    if (!misc) return;
    __syncthreads();
    float m = 3.0e38f;
    for (int i = threadIdx.x; i < B; i += 256) m = fminf(m, Z[i]);
    s[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] = fminf(s[threadIdx.x], s[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) misc[0] = s[0];
}

// ---------------------------------------------------------------------------------------
// backward prep: V[b,:] = (g / (temp * Zb')) * e1^_b   with e1^ = E1s * temp/log2e
//   => V = E1s * g * ln2 / Zb'
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void infonce_make_v_kernel(const float *E1s, const float *Z, const float *gscale,
                                                             int B, int d, int variant, float *V) {
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    const float g = gscale[0];
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        const float zb = Z[b] + (variant == 0 ? 0.f : 1e-8f);
        const float c = g * LN2_F / zb;
        for (int k = lane; k < d; k += 64) V[(size_t)b * d + k] = E1s[(size_t)b * d + k] * c;
    }
}

// ---------------------------------------------------------------------------------------
// backward hot kernel 1: Wpart[split][b,:] = sum_{j in split} exp2(<e1s_b,a_j>) a_j
// ---------------------------------------------------------------------------------------
// ZSUM: also emit zpart[split][b] = sum_{j in split} exp2(.) -- the forward pass of a call that will be differentiated runs this
// kernel instead of the row-sum kernel (SSLREC_INFONCE_FWD_W), so the backward pass does not recompute the scores for W
template <int D, bool ZSUM>
__global__ __launch_bounds__(256, 1) void infonce_bwd_anchor_kernel(const float *__restrict__ E1s,
                                                                    const float *__restrict__ An, int B, int M,
                                                                    int n_agroup, int cols_per_split,
                                                                    float *__restrict__ Wpart, float *__restrict__ zpart) {
    constexpr int HALF = IC<D>::HALF, TA = IC<D>::TA, ROWS = IC<D>::ROWS, NDT = IC<D>::NDT;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int ag = blockIdx.x % n_agroup;
    const int split = blockIdx.x / n_agroup;
    const int a0 = (ag * 4 + wave_in_block()) * ROWS;
    if (a0 >= B) return;
    const int j_begin = split * cols_per_split;
    int j_end = j_begin + cols_per_split;
    if (j_end > M) j_end = M;

    float e1[TA][HALF];
#pragma unroll
    for (int t = 0; t < TA; ++t) load_frag<D>(e1[t], E1s, a0 + t * 32, B, lane);
    f32x16 wacc[TA][NDT];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) wacc[t][dt] = zero16();
    float rs[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) rs[t] = 0.f;

    float an[HALF];
    f32x16 s_cur = zero16();
    if (j_begin < j_end) {
        load_frag<D>(an, An, j_begin, M, lane);
        s_cur = tile_dot<HALF>(an, e1[0]);
    }
    for (int j0 = j_begin; j0 < j_end; j0 += 32) {
        float at[NDT][16];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) load_frag_t<D>(at[dt], An, j0, j_end, dt, lane);   // rows >= j_end -> 0
        float nx[HALF];
        const bool more = (j0 + 32 < j_end);
        if (more) load_frag<D>(nx, An, j0 + 32, M, lane);
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            // next score tile's MFMA chain first: it does not depend on the exp2 of the current tile
            f32x16 s_nxt = s_cur;
            if (t + 1 < TA) {
                s_nxt = tile_dot<HALF>(an, e1[t + 1]);
            } else if (more) {
                s_nxt = tile_dot<HALF>(nx, e1[0]);
            }
            const f32x16 s = s_cur;                             // s[j][anchor]
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[r]);
            if constexpr (ZSUM) {              // rows >= j_end (clamped copies in the row-major fragment) stay out of the row sum
                float part = 0.f;
                if (j0 + 32 <= j_end) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part += p[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part += (j0 + crow(r, h) < j_end) ? p[r] : 0.f;
                }
                rs[t] += part;
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r)   // W^T[dd][anchor] += A^T[dd][j] * P^T[j][anchor]
                    wacc[t][dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[dt][r], p[r], wacc[t][dt], 0, 0, 0);
            s_cur = s_nxt;
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < HALF; ++k) an[k] = nx[k];
        }
    }
    // C layout of wacc: col = anchor (lane&31), row = dd = crow(r,h): 4 consecutive dd per (r>>2)
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const int b = a0 + t * 32 + (lane & 31);
        if constexpr (ZSUM) {
            const float tot = rs[t] + __shfl_xor(rs[t], 32, 64);
            if (h == 0 && b < B) zpart[(size_t)split * B + b] = tot;
        }
        if (b < B) {
            float *dst = Wpart + ((size_t)split * B + b) * D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
                    v[0] = wacc[t][dt][4 * q + 0];
                    v[1] = wacc[t][dt][4 * q + 1];
                    v[2] = wacc[t][dt][4 * q + 2];
                    v[3] = wacc[t][dt][4 * q + 3];
                    *reinterpret_cast<f32x4 *>(dst + dt * 32 + 8 * q + 4 * h) = v;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward hot kernel 2: dA[j,:] = sum_b exp2(<e1s_b,a_j>) V[b,:]   (wave owns TJ*32 rows j)
// ---------------------------------------------------------------------------------------
template <int D, int TJ>
__global__ __launch_bounds__(256, 1) void infonce_bwd_all_kernel(const float *__restrict__ E1s,
                                                                 const float *__restrict__ V,
                                                                 const float *__restrict__ An, int B, int M,
                                                                 float *__restrict__ dA) {
    constexpr int HALF = IC<D>::HALF, NDT = IC<D>::NDT;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int j0 = (blockIdx.x * 4 + wave_in_block()) * (TJ * 32);
    if (j0 >= M) return;

    float an[TJ][HALF];
#pragma unroll
    for (int t = 0; t < TJ; ++t) load_frag<D>(an[t], An, j0 + t * 32, M, lane);
    f32x16 acc[TJ][NDT];
#pragma unroll
    for (int t = 0; t < TJ; ++t)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) acc[t][dt] = zero16();

    float e1[HALF];
    load_frag<D>(e1, E1s, 0, B, lane);
    f32x16 s_cur = tile_dot<HALF>(e1, an[0]);
    for (int b0 = 0; b0 < B; b0 += 32) {
        float vt[NDT][16];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) load_frag_t<D>(vt[dt], V, b0, B, dt, lane);   // anchors >= B -> 0
        float nx[HALF];
        const bool more = (b0 + 32 < B);
        if (more) load_frag<D>(nx, E1s, b0 + 32, B, lane);
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            f32x16 s_nxt = s_cur;
            if (t + 1 < TJ) {
                s_nxt = tile_dot<HALF>(e1, an[t + 1]);
            } else if (more) {
                s_nxt = tile_dot<HALF>(nx, an[0]);
            }
            const f32x16 s = s_cur;                             // s[anchor][j]
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r)   // dA^T[dd][j] += V^T[dd][anchor] * P[anchor][j]
                    acc[t][dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[dt][r], p[r], acc[t][dt], 0, 0, 0);
            s_cur = s_nxt;
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < HALF; ++k) e1[k] = nx[k];
        }
    }
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int j = j0 + t * 32 + (lane & 31);
        if (j < M) {
            float *dst = dA + (size_t)j * D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
                    v[0] = acc[t][dt][4 * q + 0];
                    v[1] = acc[t][dt][4 * q + 1];
                    v[2] = acc[t][dt][4 * q + 2];
                    v[3] = acc[t][dt][4 * q + 3];
                    *reinterpret_cast<f32x4 *>(dst + dt * 32 + 8 * q + 4 * h) = v;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward finish for the anchor side: one wavefront per anchor
//   du1 = -g*mask*e2^/temp + (g/(temp*Z')) * sum_split Wpart ;  du2 = -g*mask*e1^/temp
//   variant 0: chain through x^ = x*rn:  dx = rn * (du - x^ <x^,du>)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void infonce_finish_bwd_kernel(const float *E1s, const float *E2n,
                                                                 const float *rn1, const float *rn2,
                                                                 const float *Wpart, int n_split, const float *Z,
                                                                 const float *gscale, int B, int d, float temp,
                                                                 int variant, float *dE1, float *dE2,
                                                                 int do_insert, const int64_t *i1, const int64_t *i2, float *dT1, float *dT2, DetTable tab) {
    // (round 6) registers the 2B gradient rows for the deterministic scatter here instead of in a launch of its own
    // (rounds 1-5: infonce_scatter_insert_kernel): the registration depends on the indices only, not on the rows this kernel is about to write;
    // row b -> dT1 + i1[b] * d, row B + b -> dT2 + i2[b] * d; a role without an index array is stored by the caller (marked -1)
    if (do_insert) {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * B; e += gridDim.x * 256) {
            const int b = e < B ? e : e - B;
            const int64_t *idx = e < B ? i1 : i2;
            float *dst = e < B ? dT1 : dT2;
            if (idx && dst) det_insert(tab, dst + idx[b] * d, e);
            else tab.slot_of[e] = -1;
        }
    }
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    const float g = gscale[0];
    const float unscale = temp * LN2_F;   // E1s * unscale = e1^
    for (int b = blockIdx.x * 4 + w; b < B; b += gridDim.x * 4) {
        const float zb = Z[b] + (variant == 0 ? 0.f : 1e-8f);
        const float cw = g / (temp * zb);
        float mask = 1.f;
        if (variant != 0) {
            float dot = 0.f;
            for (int k = lane; k < d; k += 64) dot = fmaf(E1s[(size_t)b * d + k], E2n[(size_t)b * d + k], dot);
            const float pos = wave_sum(dot) * LN2_F;
            mask = (pos >= -5.f && pos <= 5.f) ? 1.f : 0.f;
        }
        const float cp = -g * mask / temp;
        // d <= 128 here: at most 2 elements per lane, statically indexed (no scratch)
        float u1[2], u2[2], d1[2], d2[2];
        float dot1 = 0.f, dot2 = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = lane + 64 * c;
            u1[c] = u2[c] = d1[c] = d2[c] = 0.f;
            if (k < d) {
                const float x1 = E1s[(size_t)b * d + k] * unscale;
                const float x2 = E2n[(size_t)b * d + k];
                float wsum = 0.f;
                for (int s = 0; s < n_split; ++s) wsum += Wpart[((size_t)s * B + b) * d + k];
                const float g1 = cp * x2 + cw * wsum;
                const float g2 = cp * x1;
                u1[c] = x1; u2[c] = x2; d1[c] = g1; d2[c] = g2;
                dot1 = fmaf(x1, g1, dot1);
                dot2 = fmaf(x2, g2, dot2);
            }
        }
        float r1 = 1.f, r2 = 1.f;
        if (variant == 0) {
            dot1 = wave_sum(dot1);
            dot2 = wave_sum(dot2);
            r1 = rn1[b];
            r2 = rn2[b];
        } else {
            dot1 = 0.f;
            dot2 = 0.f;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = lane + 64 * c;
            if (k < d) {
                dE1[(size_t)b * d + k] = r1 * (d1[c] - u1[c] * dot1);
                dE2[(size_t)b * d + k] = r2 * (d2[c] - u2[c] * dot2);
            }
        }
    }
}

// dALL[j,:] = rn_j * (dA_j - a^_j <a^_j, dA_j>)   (in place on dA), dA_j = sum over the n_split anchor splits of the `all`-gradient
// role's partials when that role was split (slab [n_split][n][d], added in split order), else what the role wrote into dA itself;
// do_norm = 0 (variant 1: no normalization): just the sum of the splits
__global__ __launch_bounds__(256) void norm_bwd_rows_kernel(const float *An, const float *rn, int n, int d, float *dA, const float *slab,
                                                            int n_split, int do_norm) {
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    for (int r = blockIdx.x * 4 + w; r < n; r += gridDim.x * 4) {
        float v[2] = {0.f, 0.f};              // d <= 128: at most two elements per lane
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = lane + 64 * c;
            if (k < d) {
                if (slab) {
                    for (int sp = 0; sp < n_split; ++sp) v[c] += slab[((size_t)sp * n + r) * d + k];
                } else {
                    v[c] = dA[(size_t)r * d + k];
                }
                if (do_norm) dot = fmaf(An[(size_t)r * d + k], v[c], dot);
            }
        }
        float inv = 1.f;
        if (do_norm) {
            dot = wave_sum(dot);
            inv = rn[r];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = lane + 64 * c;
            if (k < d) dA[(size_t)r * d + k] = do_norm ? inv * (v[c] - An[(size_t)r * d + k] * dot) : v[c];
        }
    }
}

#include "infonce_x3.inc"

// ---------------------------------------------------------------------------------------
// Folded preparation / finishing kernels: a call spends ~0.12 of its 1.26 ms (cfg-3 item term, x6) in ~15 small launches
// around the three hot kernels; these do the same work in 5.
// ---------------------------------------------------------------------------------------
// the three row sets of a call (`all`, anchors, positives) in ONE launch; in the split-precision modes the row-major bf16
// planes of `all` and of the anchors are written here too (the split_rm passes re-read what this kernel had in registers)
struct PrepSet {
    const float *src; const int64_t *idx; int n; float scale; float *dst; float *rn;
    u16 *p0, *p1, *p2;
    float pscale;      // h3: the planes are the (hi, lo) fp16 pair of pscale * row (p2 unused); 0 = three bf16 planes
    const float *pscale_dev;   // (h3 on un-normalized rows) the scale chosen on the device from the set's largest magnitude; null: pscale
};
struct PrepArgs { PrepSet s[3]; int d, do_norm; };

__global__ __launch_bounds__(256) void prep_rows3_kernel(PrepArgs a) {
    const int lane = threadIdx.x & 63;
    const int w = wave_in_block();
    const int n01 = a.s[0].n + a.s[1].n, total = n01 + a.s[2].n;
    const int d = a.d;
    for (int rr = blockIdx.x * 4 + w; rr < total; rr += gridDim.x * 4) {
        const int which = rr < a.s[0].n ? 0 : (rr < n01 ? 1 : 2);
        const PrepSet &q = a.s[which];
        const int r = rr - (which == 0 ? 0 : (which == 1 ? a.s[0].n : n01));
        const float *x = q.src + (q.idx ? q.idx[r] : (int64_t)r) * d;
        float inv = 1.f;
        if (a.do_norm) {
            float ss = 0.f;
            for (int k = lane; k < d; k += 64) ss = fmaf(x[k], x[k], ss);
            ss = wave_sum(ss);
            inv = 1.f / sqrtf(1e-8f + ss);
        }
        for (int k = lane; k < d; k += 64) {
            const float v = (x[k] * inv) * q.scale;
            const size_t at = (size_t)r * d + k;
            q.dst[at] = v;
            if (q.p0 && q.pscale != 0.f) {
                u16 hi, lo;
                f16_split(v * q.pscale, hi, lo);
                q.p0[at] = hi;
                q.p1[at] = lo;
            } else if (q.p0) {                            // x = a + b + c in three bf16 planes
                const float hi = bf16_val(v);
                const float r1 = v - hi;
                const float mid = bf16_val(r1);
                q.p0[at] = bf16_bits(hi);
                q.p1[at] = bf16_bits(mid);
                q.p2[at] = bf16_bits(r1 - mid);
            }
        }
        if (lane == 0 && q.rn) q.rn[r] = inv;
    }
}

// Round 6: prep_rows3 + split_tt in ONE launch (split-precision modes).  A workgroup takes a 32-row tile of one of the three row sets:
// every wave normalizes 8 rows (fp32 copy, 1/|row|, row-major planes -- as prep_rows3_kernel), the scaled rows are left in LDS and the
// 256 threads then write the tile's TILE-TRANSPOSED planes (the layout of split_tt_kernel: one 16-byte store per lane), which the
// separate pass had to re-read from HBM.  Rows past the set's end contribute zeros to the transposed planes.  The first thread also
// zeroes the tickets of the one-launch forward finish (FinWs).
struct PrepTileArgs {
    PrepSet s[3];
    u16 *tt0[3], *tt1[3], *tt2[3];     // tile-transposed planes per set (null: none)
    int d, do_norm;
    int tiles0, tiles01;               // tiles of set 0, of sets 0 + 1
    unsigned *tickets; int n_tickets;  // zeroed here (stride 32 floats)
};

template <int D>
__global__ __launch_bounds__(256) void prep_tiles_kernel(PrepTileArgs a) {
    __shared__ float tile[32][D + 1];
    const int lane = threadIdx.x & 63, w = wave_in_block();
    if (blockIdx.x == 0 && (int)threadIdx.x < a.n_tickets) a.tickets[32 * threadIdx.x] = 0u;
    const int which = (int)blockIdx.x < a.tiles0 ? 0 : ((int)blockIdx.x < a.tiles01 ? 1 : 2);
    const PrepSet &q = a.s[which];
    const int T = (int)blockIdx.x - (which == 0 ? 0 : (which == 1 ? a.tiles0 : a.tiles01));
    const bool want_tt = a.tt0[which] != nullptr;
    const float pscale = q.pscale_dev ? q.pscale_dev[0] : q.pscale;
    const float ps = pscale != 0.f ? pscale : 1.f;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int lr = w * 8 + rr, r = T * 32 + lr;
        if (r < q.n) {
            const float *x = q.src + (q.idx ? q.idx[r] : (int64_t)r) * D;
            float xv[(D + 63) / 64];
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < (D + 63) / 64; ++c) {
                const int k = lane + 64 * c;
                xv[c] = k < D ? x[k] : 0.f;
                ss = fmaf(xv[c], xv[c], ss);
            }
            float inv = 1.f;
            if (a.do_norm) {
                ss = wave_sum(ss);
                inv = 1.f / sqrtf(1e-8f + ss);
            }
#pragma unroll
            for (int c = 0; c < (D + 63) / 64; ++c) {
                const int k = lane + 64 * c;
                if (k < D) {
                    const float v = (xv[c] * inv) * q.scale;              // same expression as prep_rows3_kernel
                    const size_t at = (size_t)r * D + k;
                    q.dst[at] = v;
                    if (q.p0 && pscale != 0.f) {
                        u16 hi, lo;
                        f16_split(v * pscale, hi, lo);
                        q.p0[at] = hi;
                        q.p1[at] = lo;
                    } else if (q.p0) {
                        const float hi = bf16_val(v);
                        const float r1 = v - hi;
                        const float mid = bf16_val(r1);
                        q.p0[at] = bf16_bits(hi);
                        q.p1[at] = bf16_bits(mid);
                        q.p2[at] = bf16_bits(r1 - mid);
                    }
                    if (want_tt) tile[lr][k] = v * ps;
                }
            }
            if (lane == 0 && q.rn) q.rn[r] = inv;
        } else if (want_tt) {
#pragma unroll
            for (int c = 0; c < (D + 63) / 64; ++c) {
                const int k = lane + 64 * c;
                if (k < D) tile[lr][k] = 0.f;
            }
        }
    }
    if (!want_tt) return;                  // (uniform per workgroup)
    __syncthreads();
    constexpr int NDT = D / 32, CHUNKS = NDT * 2 * 2 * 32;      // 16-byte chunks of one plane of the tile
    const bool f16 = pscale != 0.f;
    for (int o8 = threadIdx.x; o8 < CHUNKS; o8 += 256) {
        const int c = o8 & 31, hh = (o8 >> 5) & 1, qq = (o8 >> 6) & 1, dt = o8 >> 7;
        u16x8 v0, v1, v2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = tile[crow(8 * qq + i, hh)][dt * 32 + c];
            if (f16) {
                u16 hi, lo;
                f16_split(x, hi, lo);
                v0[i] = hi; v1[i] = lo; v2[i] = 0;
            } else {
                const float b0 = bf16_val(x);
                const float r1 = x - b0;
                const float b1 = bf16_val(r1);
                v0[i] = bf16_bits(b0); v1[i] = bf16_bits(b1); v2[i] = bf16_bits(r1 - b1);
            }
        }
        const size_t at = ((size_t)T * CHUNKS + o8) * 8;
        *reinterpret_cast<u16x8 *>(a.tt0[which] + at) = v0;
        *reinterpret_cast<u16x8 *>(a.tt1[which] + at) = v1;
        if (!f16) *reinterpret_cast<u16x8 *>(a.tt2[which] + at) = v2;
    }
}

// ---------------------------------------------------------------------------------------
// h3 on the UN-NORMALIZED variant (LightGCL, lightgcl.py:114-118; round 6).  The normalized variant's fp16 scales are constants
// (|a| <= 1, |e1s| <= log2e / temp); here rows and scores are unbounded, so three things are chosen on the device per call:
//   * the plane scales 2^ka (`all`) and 2^ke (anchors) from the tables' largest magnitudes: scaled maxima in (2^12, 2^13];
//   * the exponent bias PER ANCHOR from the anchor's largest score (row-max pre-pass on the high planes): P' = exp2(t - ceil(max t) + 13)
//     stays inside fp16 whatever the scores are, and the entries that matter (within 2^-24 of the anchor's largest) keep 22 bits;
//   * the V scale from min_b (Z_b + 1e-8) 2^bias_b (forward finish), as in the normalized variant.
// Row sums and W go back to their TRUE scale when written (x 2^-bias_b): Z_b = sum_j exp(s_bj) overflows fp32 exactly where the reference's
// own exp() does (s > 88.7) -- its value is matched, not improved on -- and everything downstream (the staged entry points that
// all-reduce Z and W across ranks included) reads the same quantities as in the other arithmetics.
// misc (floats): [0] min_b (Z_b + eps) 2^bias_b   [1] all-gradient role's output scale   [8] max |all| (bits)   [9] max |e1s| (bits)
//                [10] 2^ka   [11] 2^ke   [12] 2^-(ka+ke)   [13] 2^-ka
// ---------------------------------------------------------------------------------------
#define DYN_MAX_ALL 8
#define DYN_MAX_E1 9
#define DYN_SC_ALL 10
#define DYN_SC_E1 11
#define DYN_SC_MUL 12
#define DYN_SC_OUT 13

// largest magnitudes of `all` and of scale * T1[i1] (non-negative floats order like their bit patterns: atomicMax on the bits)
__global__ __launch_bounds__(256) void dyn_maxabs_kernel(const float *__restrict__ all, size_t n_all, const float *__restrict__ T1,
                                                         const int64_t *__restrict__ i1, int B, int d, float e1_scale, unsigned *misc_bits) {
    __shared__ float sm[2][4];
    float m0 = 0.f, m1 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_all; i += (size_t)gridDim.x * 256) m0 = fmaxf(m0, fabsf(all[i]));
    const size_t n_e = (size_t)B * d;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_e; i += (size_t)gridDim.x * 256) {
        const size_t b = i / d, k = i - b * d;
        m1 = fmaxf(m1, fabsf(T1[(i1 ? i1[b] : (int64_t)b) * d + k] * e1_scale));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o, 64)); m1 = fmaxf(m1, __shfl_xor(m1, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = m0; sm[1][threadIdx.x >> 6] = m1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(misc_bits + DYN_MAX_ALL, __float_as_uint(fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]))));
        atomicMax(misc_bits + DYN_MAX_E1, __float_as_uint(fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]))));
    }
}

__device__ __forceinline__ float dyn_pow2_scale(float maxabs) {      // 2^k with k = 13 - ceil(log2 maxabs): scaled maximum in (2^12, 2^13]
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.f;
    return exp2f(13.f - ceilf(log2f(maxabs)));
}

__global__ void dyn_scales_kernel(float *misc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float sa = dyn_pow2_scale(misc[DYN_MAX_ALL]), se = dyn_pow2_scale(misc[DYN_MAX_E1]);
    misc[DYN_SC_ALL] = sa;
    misc[DYN_SC_E1] = se;
    misc[DYN_SC_MUL] = (1.f / sa) * (1.f / se);      // (powers of two: exact)
    misc[DYN_SC_OUT] = 1.f / sa;
}

// bias[b] = 13 - ceil(largest score of anchor b) from the row-max partials (score = accumulator x sc_mul, in log2 units)
__global__ __launch_bounds__(256) void dyn_bias_kernel(const float *__restrict__ mpart, int n_split, int B, int b32, const float *misc,
                                                       float *__restrict__ bias) {
    const float sc_mul = misc[DYN_SC_MUL];
    for (int b = blockIdx.x * 256 + threadIdx.x; b < b32; b += gridDim.x * 256) {
        // rows B .. b32 of a streamed anchor tile are clamped copies of anchor B - 1 (their V planes are zero): they take ITS bias, so that
        // their P' stays finite in fp16 (inf x 0 would put NaN into the second product)
        const int bb_ = b < B ? b : B - 1;
        float m = -3.0e38f;
        for (int sp = 0; sp < n_split; ++sp) m = fmaxf(m, mpart[(size_t)sp * B + bb_]);
        // (the pre-pass sees the high planes only: +1 covers its ~2^-10 |score| error up to scores of ~500; clamped so that exp2f(+-bias)
        // stays a finite fp32 factor)
        float bb = 13.f - ceilf(m * sc_mul + 1.f);
        bb = fminf(fmaxf(bb, -110.f), 110.f);
        bias[b] = bb;
    }
}

// backward prologue in the split-precision modes: V = E1s * g ln2 / Z' is never written in fp32 -- its tile-transposed planes
// are computed directly (make_v + split_tt(V) in one launch); the first threads also clear the scatter table of the call
__global__ __launch_bounds__(256) void make_v_tt_kernel(const float *__restrict__ E1s, const float *__restrict__ Z, const float *gscale,
                                                        int n, int d, int variant, u16 *__restrict__ p0, u16 *__restrict__ p1,
                                                        u16 *__restrict__ p2, DetTable tab, int clear_tab, int f16, float smax,
                                                        float p_bias, float *__restrict__ misc, const float *__restrict__ dyn_bias) {
    if (clear_tab) det_clear_from(tab, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
    const float g = gscale[0];
    // dyn (h3, un-normalized variant): V'_b = 2^-bias_b V_b -- the all-gradient role multiplies it with P' = 2^bias_b P -- and the bound of
    // its magnitude comes from the device: max |e1s| (misc[DYN_MAX_E1]) over min_b (Z_b + eps) 2^bias_b (misc[0], forward finish)
    if (dyn_bias) { smax = misc[DYN_MAX_E1]; p_bias = 0.f; }
    // h3: |V| <= |g| ln2 smax / min_b Z_b (misc[0], left by the forward pass) -> 2^kv V stays below 2^13; the all-gradient role
    // multiplies its accumulators by misc[1] = 2^-(kv + bias)
    float vscale = 1.f;
    if (f16) {
        const float vmax = fabsf(g) * LN2_F * smax / fmaxf(misc[0], 1e-30f);
        const int kv = (vmax > 0.f && vmax < 3.0e38f) ? 13 - (int)ceilf(log2f(vmax)) : 0;
        vscale = exp2f((float)kv);
        if (blockIdx.x == 0 && threadIdx.x == 0) misc[1] = exp2f(-((float)kv + p_bias));
    }
    const int ndt = d / 32;
    const size_t total = (size_t)((n + 31) / 32) * ndt * 2 * 2 * 32 * 8;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int i = (int)(o & 7);
        const int c = (int)((o >> 3) & 31);
        const int h = (int)((o >> 8) & 1);
        const int q = (int)((o >> 9) & 1);
        const size_t rest = o >> 10;
        const int dt = (int)(rest % ndt);
        const size_t T = rest / ndt;
        const size_t row = T * 32 + crow(8 * q + i, h);
        float x = 0.f;
        if (row < (size_t)n) {
            const float zb = Z[row] + (variant == 0 ? 0.f : 1e-8f);
            x = E1s[row * d + dt * 32 + c] * (g * LN2_F / zb);      // same expression as infonce_make_v_kernel
            if (dyn_bias) x *= exp2f(-dyn_bias[row]);
        }
        if (f16) {
            u16 hi, lo;
            f16_split(x * vscale, hi, lo);
            p0[o] = hi;
            p1[o] = lo;
            continue;
        }
        const float a = bf16_val(x);
        const float r1 = x - a;
        const float b = bf16_val(r1);
        p0[o] = bf16_bits(a);
        p1[o] = bf16_bits(b);
        p2[o] = bf16_bits(r1 - b);
    }
}

__global__ __launch_bounds__(256) void det_clear_only_kernel(DetTable tab) {
    det_clear_from(tab, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// ---------------------------------------------------------------------------------------
// host side: workspace carving and launch sequencing
// ---------------------------------------------------------------------------------------
#define INF_FIN_BLOCKS 256
#define INF_CUS 256

struct InfPlan {
    int rows_per_wave, n_agroup, n_split, cols_per_split;
    int n_bsplit;      // anchor splits of the `all`-gradient role (split-precision modes): > 1 when M / 128 workgroups would not fill the chip
    size_t off_dapart; // its partial slab [n_bsplit][M][d] (n_bsplit > 1)
    size_t off_an, off_e1s, off_e2n, off_rn1, off_rn2, off_rna, off_z, off_zpart, off_part, off_misc, off_fin, off_bias, off_v, off_wpart,
        off_an_rm, off_an_tt, off_e1_rm, off_v_tt,   // bf16 planes (hi then lo), see infonce_x3.inc
        total;   // offsets in floats
};

static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

static InfPlan make_plan(int B, int M, int d) {
    InfPlan p;
    p.rows_per_wave = (d == 128 ? 2 : 4) * 32;
    const int rows_per_block = 4 * p.rows_per_wave;
    p.n_agroup = (B + rows_per_block - 1) / rows_per_block;
    if (p.n_agroup < 1) p.n_agroup = 1;
    int ns = INF_CUS / p.n_agroup;
    if (ns < 1) ns = 1;
    const int max_split = (M + 31) / 32;
    if (ns > max_split) ns = max_split;
    if (ns < 1) ns = 1;
    int cps = (M + ns - 1) / ns;
    cps = (cps + 31) / 32 * 32;
    if (cps < 32) cps = 32;
    p.cols_per_split = cps;
    p.n_split = (M + cps - 1) / cps;
    if (p.n_split < 1) p.n_split = 1;
    size_t o = 0;
    p.off_an = o;    o += align64((size_t)M * d);
    p.off_e1s = o;   o += align64((size_t)B * d);
    p.off_e2n = o;   o += align64((size_t)B * d);
    p.off_rn1 = o;   o += align64(B);
    p.off_rn2 = o;   o += align64(B);
    p.off_rna = o;   o += align64(M);
    p.off_z = o;     o += align64(B);
    p.off_zpart = o; o += align64((size_t)p.n_split * B);
    p.off_part = o;  o += align64(INF_FIN_BLOCKS);
    p.off_misc = o;  o += 64;      // [0] min_b Z_b (forward), [1] output scale of the h3 all-gradient role (backward)
    p.off_bias = o;  o += align64((size_t)(B + 31) / 32 * 32);      // (dyn) per-anchor exponent bias, padded to whole 32-row tiles
    p.off_fin = o;   o += align64(FINF_FLOATS);      // tickets + per-workgroup partials of the one-launch forward finish
    p.off_v = o;     o += align64((size_t)B * d);
    p.off_wpart = o; o += align64((size_t)p.n_split * B * d);
    const size_t m32 = (size_t)(M + 31) / 32 * 32, b32 = (size_t)(B + 31) / 32 * 32;
    p.off_an_rm = o; o += align64(((size_t)M * d * 3 + 1) / 2);          // 3 planes x M*d bf16
    p.off_an_tt = o; o += align64((m32 * d * 3 + 1) / 2);       // 3 planes
    p.off_e1_rm = o; o += align64(((size_t)B * d * 3 + 1) / 2);
    p.off_v_tt = o;  o += align64((b32 * d * 3 + 1) / 2);
    // The `all`-gradient role keeps 128 `all` rows resident per workgroup and streams every anchor tile: M / 128 workgroups.  Three fit a
    // CU (768 chip-wide); a small `all` table (yelp's 26,822 items: 210 workgroups) leaves most of the chip idle for the length of the
    // anchor stream, so the stream is cut in TWO halves whose partial results go to a slab that the row-normalization pass adds up
    // (fixed order: deterministic).  Only when both halves fit the chip together (<= 384 workgroups) and are >= 16 tiles long: measured
    // on cfg 4 (SGL-ED, real yelp), 1 / 2 / 3 / 4 parts: 1.706 / 1.625 / 1.624 / 1.656 ms per step; splitting cfg 3's user term (412
    // workgroups) did not pay (profiles/r04/infonce_bsplit.json).
    {
        const int n_rgroup = (M + 127) / 128, tiles = (B + 31) / 32;
        int nb = (2 * n_rgroup <= 3 * INF_CUS && tiles >= 32) ? 2 : 1;
        static const int forced = [] { const char *e = getenv("SSLREC_INFONCE_BSPLIT"); return e ? atoi(e) : 0; }();      // experiments: 1 = never split
        if (forced >= 1) nb = forced > tiles / 2 && tiles >= 2 ? tiles / 2 : (forced > tiles ? 1 : forced);
        p.n_bsplit = nb;
        p.off_dapart = o;
        if (nb > 1) o += align64((size_t)nb * M * d);
    }
    p.total = o;
    return p;
}

extern "C" size_t sslrec_infonce_ws_bytes(int32_t B, int32_t M, int32_t d) {
    if (B <= 0 || M <= 0 || d <= 0) return 0;
    return make_plan(B, M, d).total * sizeof(float);
}

static X3Planes x3_planes(const InfPlan &p, float *ws, int B, int M, int d) {
    const size_t m32 = (size_t)(M + 31) / 32 * 32, b32 = (size_t)(B + 31) / 32 * 32;
    X3Planes x;
    u16 *b = reinterpret_cast<u16 *>(ws + p.off_an_rm);
    for (int k = 0; k < 3; ++k) x.an_rm[k] = b + (size_t)k * M * d;
    b = reinterpret_cast<u16 *>(ws + p.off_e1_rm);
    for (int k = 0; k < 3; ++k) x.e1_rm[k] = b + (size_t)k * B * d;
    b = reinterpret_cast<u16 *>(ws + p.off_an_tt);
    for (int k = 0; k < 3; ++k) x.an_tt[k] = b + (size_t)k * m32 * d;
    b = reinterpret_cast<u16 *>(ws + p.off_v_tt);
    for (int k = 0; k < 3; ++k) x.v_tt[k] = b + (size_t)k * b32 * d;
    return x;
}

// Precision of the B x M products, SSLREC_INFONCE_PRECISION = h3 (default, variant 0) | x6 (default, variant 1) | x36 | x3 | fp32:
//   x6    3 bf16 planes / 6 terms for the scores and for the second products (P.A, P^T.V): 24-bit operands, error
//         1.6e-7 on the scores against fp64 (exact-fp32 MFMA: 2.7e-7) -- passes every parity test at the fp32 tolerances
//   fp32  exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference arithmetic itself
//   x36   opt-in: scores from 2 planes / 3 terms (error 4e-6), second products 3 planes / 6 terms; gradients within 1e-5
//         of their scale, the golden reference steps still pass at rtol 1e-4
//   x3    opt-in: 2 planes / 3 terms everywhere (fastest; ~1e-4-of-scale noise in the gradients)
// Variant 1 (LightGCL, un-normalized and therefore unbounded scores) runs x6 or fp32 only: the opt-in modes' score
// error is absolute in the operand scale.
// The caller selects the mode in bits 8..15 of `variant` (SSLREC_INFONCE_X6 ... in sslrec_hip.h; forward and backward of
// one call must pass the same value); 0 there = the process-wide default, SSLREC_INFONCE_PRECISION or x6.
//   h3    opt-in (round 5): TWO fp16 planes / 3 terms everywhere (infonce_x3.inc): 22-bit operands, half of x6's matrix instructions
struct InfPrec { int np, ns, ns_all; bool f16; bool dyn = false; };      // dyn: h3 on un-normalized rows: scales and per-anchor bias chosen on the device      // planes of the score product / of the second products (anchor-gradient role, all-gradient role); np = 0: fp32
static InfPrec inf_precision(int variant_full) {
    const int variant = variant_full & 0xFF, code = (variant_full >> 8) & 0xFF;
    static const char *const names[] = {nullptr, "x6", "fp32", "x36", "x3", "x63", "x6a", "h3"};
    const char *e = (code >= 1 && code <= 7) ? names[code] : getenv("SSLREC_INFONCE_PRECISION");
    // the default since round 5: h3 on normalized rows (errors against fp64 equal to x6's and the exact-fp32 kernels' to three digits,
    // 0.65 against 1.03 ms for cfg 3's item term: profiles/r05/infonce_modes.json; every parity test of the suite passes with it),
    // x6 for the un-normalized variant
    // round 6: the un-normalized variant has an h3 of its own (device-chosen plane scales, per-anchor exponent bias: see dyn_* above);
    // its default is SSLREC_INFONCE_V1_DEFAULT (h3 | x6)
    if (!e || !*e) {
        if (variant == 0) return InfPrec{2, 2, 2, true};
        static const bool v1_h3 = [] { const char *v = getenv("SSLREC_INFONCE_V1_DEFAULT"); return !(v && v[0] == 'x'); }();
        return v1_h3 ? InfPrec{2, 2, 2, true, true} : InfPrec{3, 3, 3, false};
    }
    if (e[0] == 'f') return {0, 0, 0, false};
    if (variant != 0 && e[0] == 'h') return {2, 2, 2, true, true};
    if (variant != 0) return {3, 3, 3, false};    // the opt-in modes' score error is absolute in the operand scale: x6 instead
    if (e[0] == 'h') return {2, 2, 2, true};
    if (e[0] == 'x' && e[1] == '6' && e[2] == 'a') return {3, 3, 2, false};   // as x6, but the all-gradient role's second product with 3 terms
    if (e[0] == 'x' && e[1] == '6' && e[2] == '3') return {3, 2, 2, false};   // scores with 6 terms, the (linear) second products with 3
    if (e[0] == 'x' && e[1] == '3' && e[2] == '6') return {2, 3, 3, false};
    if (e[0] == 'x' && e[1] == '3') return {2, 2, 2, false};
    return {3, 3, 3, false};
}

// h3: exponent bias of P' = exp2(score + bias) (infonce_x3.inc): P' < 2^15.5 for |score| <= log2e / temp
static float h3_bias(float temp) {
    const float b = floorf(15.5f - LOG2E_F / temp);
    return b < 7.f ? b : 7.f;
}

// h3 needs bias >= 0: below temp = log2e / 15.5 = 0.0931 the exponent bias goes negative, P' = exp2(score + bias) sinks towards
// fp16's subnormals and the second products (anchor gradient W, dALL) keep only a few bits.  There the call runs x6 --
// three bf16 planes have fp32's exponent range -- whether h3 was the default or asked for by name; forward and backward of a call
// receive the same temp, so they resolve alike.
static int inf_resolve(int variant_full, float temp) {
    if (!inf_precision(variant_full).f16 || inf_precision(variant_full).dyn || h3_bias(temp) >= 0.f) return variant_full;
    return (variant_full & ~0xFF00) | (SSLREC_INFONCE_PREC_X6 << 8);
}

static int grid_for_elems_x3(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

static int split_tt(const float *src, int n, int d, const u16 *const (&pl)[3], hipStream_t st, bool f16 = false, float scale = 1.f,
                    const float *scale_dev = nullptr) {
    const size_t total = (size_t)((n + 31) / 32) * 32 * d;
    hipLaunchKernelGGL(split_tt_kernel, dim3(grid_for_elems_x3(total)), dim3(256), 0, st, src, n, d, const_cast<u16 *>(pl[0]),
                       const_cast<u16 *>(pl[1]), const_cast<u16 *>(pl[2]), f16 ? 1 : 0, scale, scale_dev);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

static bool inf_variant_ok(int variant) {
    const int prec = (variant >> 8) & 0xFF;
    return ((variant & 0xFF) == 0 || (variant & 0xFF) == 1) && prec <= 7 && (variant & ~(0xFFFF | SSLREC_INFONCE_FWD_W)) == 0;
}

static bool inf_args_ok(const float *T1, const float *T2, int B, const float *ALL, int M, int d, float temp,
                        int variant) {
    return T1 && T2 && ALL && B > 0 && M > 0 && (d == 32 || d == 64 || d == 128) && temp > 0.f && inf_variant_ok(variant);
}

static int grid_for_rows(int n) {
    int b = (n + 3) / 4;
    return b > 2048 ? 2048 : (b < 1 ? 1 : b);
}

template <int D>
static int launch_rowsum(const InfPlan &p, const float *E1s, const float *An, int B, int M, float *zpart,
                         hipStream_t st) {
    constexpr int TA_F = (D <= 64) ? 2 : IC<D>::TA;          // forward: 2 resident anchor tiles, 2 waves/SIMD
    const int n_agroup_f = (B + 4 * TA_F * 32 - 1) / (4 * TA_F * 32);
    hipLaunchKernelGGL((infonce_rowsum_kernel<D, TA_F>), dim3(n_agroup_f * p.n_split), dim3(256), 0, st, E1s, An, B,
                       M, n_agroup_f, p.cols_per_split, zpart);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// resident 32-row tiles per wave of the split-precision kernels: 3 score planes take 1.5x the registers
template <int D, int NP> struct XT { static constexpr int TA = IC<D>::TA; };

template <int D, int NP, bool F16 = false>
static int launch_rowsum_x3(const InfPlan &p, const X3Planes &x, int B, int M, float *zpart, hipStream_t st, float temp = 1.f,
                            const float *dyn_sc = nullptr, const float *dyn_bias = nullptr) {
    // one wave per SIMD with IC<D>::TA resident anchor tiles; two waves per SIMD with half the tiles (all operands in
    // VGPRs, no AGPR shuffling) measured 9 % SLOWER: the streamed operand is then fetched twice as often
    constexpr int TA = XT<D, NP>::TA;
    const int n_agroup = (B + 4 * TA * 32 - 1) / (4 * TA * 32);
    const float bias = F16 ? h3_bias(temp) : 0.f;
    hipLaunchKernelGGL((infonce_rowsum_x3_kernel<D, TA, NP, 1, F16>), dim3(n_agroup * p.n_split), dim3(256), 0, st, x, B, M,
                       n_agroup, p.cols_per_split, zpart, H3_SCORE_UNSCALE, bias, exp2f(-bias), dyn_sc, dyn_bias);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// (dyn) row-max pre-pass + per-anchor bias: mpart aliases zpart (consumed before the hot pass writes its row sums there)
template <int D>
static int launch_dyn_bias(const InfPlan &p, const X3Planes &x, float *ws, int B, int M, hipStream_t st) {
    constexpr int TA = IC<D>::TA;
    const int n_agroup = (B + 4 * TA * 32 - 1) / (4 * TA * 32);
    float *mpart = ws + p.off_zpart;
    hipLaunchKernelGGL((infonce_rowmax_h_kernel<D, TA>), dim3(n_agroup * p.n_split), dim3(256), 0, st, x, B, M, n_agroup, p.cols_per_split, mpart);
    SSLREC_LAUNCH_CHECK();
    const int b32 = (B + 31) / 32 * 32;
    hipLaunchKernelGGL(dyn_bias_kernel, dim3((b32 + 255) / 256), dim3(256), 0, st, mpart, p.n_split, B, b32, ws + p.off_misc, ws + p.off_bias);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// SSLREC_INFONCE_PIPE=1: the software-pipelined loop (infonce_x3.inc, PIPE: the scores of tile T + 1 issued before the vector work
// and the second product of tile T; three stage buffers).  Measured in round 6 on cfg 3's item term (profiles/r06/infonce_pipe_ab.json):
// h3 forward + backward 0.6186 ms against 0.6185 ms for the plain order of work, x6 1.055 against 0.973 -- with two or three waves per
// SIMD the other waves already fill a wave's matrix-only and vector-only stretches, and the third buffer costs x6 a workgroup of LDS.
// NEGATIVE: off by default, kept for the record and for other shapes.
static bool inf_pipe() {
    static const bool on = [] { const char *e = getenv("SSLREC_INFONCE_PIPE"); return e && e[0] == '1'; }();
    return on;
}

template <int D, int TR, int NP, int NS, bool ZSUM, bool F16, bool PIPE, bool DYN = false>
static int launch_bwd_lds_p(const LdsBwdArgs &a, int n_blocks, hipStream_t st) {
    typedef StageGeom<D, NP, NS> SG;
    const size_t lds = (size_t)(PIPE ? 3 : 2) * SG::SLABS * 1024;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SSLREC_E_BADARG;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)infonce_bwd_lds_kernel<D, TR, NP, NS, ZSUM, F16, PIPE, DYN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((infonce_bwd_lds_kernel<D, TR, NP, NS, ZSUM, F16, PIPE, DYN>), dim3(n_blocks), dim3(256), lds, st, a);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

template <int D, int TR, int NP, int NS, bool ZSUM = false, bool F16 = false>
static int launch_bwd_lds(const LdsBwdArgs &a, int n_blocks, hipStream_t st) {
    // three stage buffers of the pipelined loop must leave the kernel its waves per SIMD (launch bounds): 160 KiB of LDS per CU
    typedef StageGeom<D, NP, NS> SG;
    constexpr int WGS = (D == 128) ? 1 : (TR == 2 && (D == 64 || ZSUM)) ? 2 : 3;
    constexpr bool FITS = 3 * SG::SLABS * WGS <= 160;
    if constexpr (FITS) {
        if (inf_pipe() && !a.dyn_sc) return launch_bwd_lds_p<D, TR, NP, NS, ZSUM, F16, true>(a, n_blocks, st);
    }
    if constexpr (F16 && NP == 2 && NS == 2) {      // (dyn: per-anchor bias; the plain loop only)
        if (a.dyn_sc) return launch_bwd_lds_p<D, TR, NP, NS, ZSUM, F16, false, true>(a, n_blocks, st);
    }
    return launch_bwd_lds_p<D, TR, NP, NS, ZSUM, F16, false>(a, n_blocks, st);
}

// backward of the split-precision modes: ONE kernel template (infonce_x3.inc, infonce_bwd_lds_kernel) in two roles
template <int D, int NP, int NS, bool ZSUM = false, bool F16 = false>
static int launch_bwd_anchor_x3(const InfPlan &p, const X3Planes &x, int B, int M, float *Wpart, float *zpart, hipStream_t st, float temp = 1.f,
                                const float *dyn_sc = nullptr, const float *dyn_bias = nullptr) {
    // resident: the anchors (64 per wave, 32 at d = 128); streamed: this column split's `all` tiles
    LdsBwdArgs a = {};
    if (F16) {      // W = sum_j P a_j: P' = 2^bias P, a' = 2^8 a
        const float bias = h3_bias(temp);
        a.sc_mul = H3_SCORE_UNSCALE; a.p_bias = bias; a.z_mul = exp2f(-bias); a.out_mul = exp2f(-bias) / H3_ALL_SCALE;
    }
    for (int k = 0; k < 3; ++k) { a.res_rm[k] = x.e1_rm[k]; a.str_rm[k] = x.an_rm[k]; a.str_tt[k] = x.an_tt[k]; }
    a.n_res = B; a.n_str = M;
    constexpr int TR = (D == 128) ? 1 : 2;      // d = 128: one tile per wave fits the register budget
    a.n_rgroup = (B + 4 * TR * 32 - 1) / (4 * TR * 32);
    a.tiles_per_split = p.cols_per_split / 32;
    a.out = Wpart;
    a.zpart = zpart;
    if (dyn_sc) { a.dyn_sc = dyn_sc + DYN_SC_MUL; a.bias_res = dyn_bias; }      // dyn_sc[0] = score scale, [1] = 2^-ka (misc[12], misc[13])
    return launch_bwd_lds<D, TR, NP, NS, ZSUM, F16>(a, a.n_rgroup * p.n_split, st);
}

template <int D, int NP, int NS, bool F16 = false>
static int launch_bwd_all_x3(const X3Planes &x, int B, int M, float *dA, int n_bsplit, hipStream_t st, float temp = 1.f,
                             const float *out_mul_dev = nullptr, const float *norm_an = nullptr, const float *norm_rn = nullptr,
                             const float *dyn_sc = nullptr, const float *dyn_bias = nullptr) {
    // resident: 32 `all` rows per wave (128 per workgroup: ~3 workgroups per CU keep the chip balanced);
    // streamed: every anchor tile (scores) with the matching rows of V (second product)
    LdsBwdArgs a = {};
    if (F16) {      // dA = sum_b P V_b: P' = 2^bias P, V' = 2^kv V with kv chosen on the device (make_v_tt_kernel -> *out_mul_dev)
        a.sc_mul = H3_SCORE_UNSCALE; a.p_bias = h3_bias(temp); a.z_mul = 1.f; a.out_mul = 1.f; a.out_mul_dev = out_mul_dev;
    }
    for (int k = 0; k < 3; ++k) { a.res_rm[k] = x.an_rm[k]; a.str_rm[k] = x.e1_rm[k]; a.str_tt[k] = x.v_tt[k]; }
    a.n_res = M; a.n_str = B;
    a.n_rgroup = (M + 127) / 128;
    a.tiles_per_split = ((B + 31) / 32 + n_bsplit - 1) / n_bsplit;      // (n_bsplit > 1: dA is the slab [n_bsplit][M][D])
    a.out = dA;
    a.zpart = nullptr;
    a.norm_an = norm_an; a.norm_rn = norm_rn;
    if (dyn_sc) { a.dyn_sc = dyn_sc + DYN_SC_MUL; a.bias_str = dyn_bias; }
    return launch_bwd_lds<D, 1, NP, NS, false, F16>(a, a.n_rgroup * n_bsplit, st);
}

template <int D, bool ZSUM = false>
static int launch_bwd_anchor(const InfPlan &p, const float *E1s, const float *An, int B, int M, float *Wpart, float *zpart,
                             hipStream_t st) {
    hipLaunchKernelGGL((infonce_bwd_anchor_kernel<D, ZSUM>), dim3(p.n_agroup * p.n_split), dim3(256), 0, st, E1s, An, B,
                       M, p.n_agroup, p.cols_per_split, Wpart, zpart);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

template <int D, int TJ>
static int launch_bwd_all_tj(const float *E1s, const float *V, const float *An, int B, int M, float *dA,
                             hipStream_t st) {
    const int waves = (M + TJ * 32 - 1) / (TJ * 32);
    hipLaunchKernelGGL((infonce_bwd_all_kernel<D, TJ>), dim3((waves + 3) / 4), dim3(256), 0, st, E1s, V, An, B, M,
                       dA);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

template <int D>
static int launch_bwd_all(const float *E1s, const float *V, const float *An, int B, int M, float *dA,
                          hipStream_t st) {
    // pick the number of resident 32-row tiles per wave so that the 1024 wave slots
    // (256 CUs x 4 SIMDs, one wave each) are filled in as few, as full rounds as possible
    constexpr int TMAX = IC<D>::TA;
    int best = 1;
    long best_cost = -1;
    for (int tj = TMAX; tj >= 1; --tj) {
        const long waves = (M + tj * 32 - 1) / (tj * 32);
        const long rounds = (waves + 4 * INF_CUS - 1) / (4 * INF_CUS);
        const long cost = rounds * (tj * 8 + 1);   // +1: per-round cost of streaming the anchor tiles
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = tj;
        }
    }
    if (best == 1) return launch_bwd_all_tj<D, 1>(E1s, V, An, B, M, dA, st);
    if (best == 2) return launch_bwd_all_tj<D, 2>(E1s, V, An, B, M, dA, st);
    if constexpr (TMAX >= 4) {
        if (best == 3) return launch_bwd_all_tj<D, 3>(E1s, V, An, B, M, dA, st);
        return launch_bwd_all_tj<D, 4>(E1s, V, An, B, M, dA, st);
    }
    return launch_bwd_all_tj<D, 2>(E1s, V, An, B, M, dA, st);
}

// hot stages shared by the single-call and the staged (row-sharded) entry points
#define SSLREC_BY_D(CALL32, CALL64, CALL128) (d == 32 ? (CALL32) : d == 64 ? (CALL64) : (CALL128))

// the three row sets of a call prepared by one launch (prep_rows3_kernel); in the split-precision modes it also writes the
// row-major planes the score products read
static int prep_all(const InfPlan &p, float *ws, const float *T1, const int64_t *i1, const float *T2, const int64_t *i2, int B,
                    const float *ALL, int M, int d, float temp, int variant_full, hipStream_t st, bool want_all_tt = false) {
    const int do_norm = ((variant_full & 0xFF) == 0);
    const bool planes = inf_precision(variant_full).np != 0;
    const X3Planes x = x3_planes(p, ws, B, M, d);
    PrepArgs a = {};
    a.d = d; a.do_norm = do_norm;
    a.s[0] = PrepSet{ALL, nullptr, M, 1.f, ws + p.off_an, ws + p.off_rna, nullptr, nullptr, nullptr, 0.f};
    a.s[1] = PrepSet{T1, i1, B, LOG2E_F / temp, ws + p.off_e1s, ws + p.off_rn1, nullptr, nullptr, nullptr, 0.f};
    a.s[2] = PrepSet{T2, i2, B, 1.f, ws + p.off_e2n, ws + p.off_rn2, nullptr, nullptr, nullptr, 0.f};
    if (inf_precision(variant_full).f16) { a.s[0].pscale = H3_ALL_SCALE; a.s[1].pscale = H3_E1_SCALE; }
    if (inf_precision(variant_full).dyn) {
        // un-normalized rows: the plane scales follow the tables' largest magnitudes (two small launches before the preparation)
        float *misc = ws + p.off_misc;
        hipError_t e = hipMemsetAsync(misc + DYN_MAX_ALL, 0, 2 * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
        const size_t n_all = (size_t)M * d;
        hipLaunchKernelGGL(dyn_maxabs_kernel, dim3((int)std::min<size_t>((n_all + 255) / 256, 2048)), dim3(256), 0, st, ALL, n_all, T1, i1, B, d,
                           LOG2E_F / temp, reinterpret_cast<unsigned *>(misc));
        SSLREC_LAUNCH_CHECK();
        hipLaunchKernelGGL(dyn_scales_kernel, dim3(1), dim3(64), 0, st, misc);
        SSLREC_LAUNCH_CHECK();
        a.s[0].pscale_dev = misc + DYN_SC_ALL;
        a.s[1].pscale_dev = misc + DYN_SC_E1;
    }
    if (planes) {
        a.s[0].p0 = const_cast<u16 *>(x.an_rm[0]); a.s[0].p1 = const_cast<u16 *>(x.an_rm[1]); a.s[0].p2 = const_cast<u16 *>(x.an_rm[2]);
        a.s[1].p0 = const_cast<u16 *>(x.e1_rm[0]); a.s[1].p1 = const_cast<u16 *>(x.e1_rm[1]); a.s[1].p2 = const_cast<u16 *>(x.e1_rm[2]);
        // round 6: one launch for the rows, their row-major planes AND (want_all_tt: the anchor-gradient role follows) the tile-transposed
        // planes of `all`; it also zeroes the tickets of the one-launch finish
        PrepTileArgs t = {};
        for (int k = 0; k < 3; ++k) t.s[k] = a.s[k];
        if (want_all_tt) { t.tt0[0] = const_cast<u16 *>(x.an_tt[0]); t.tt1[0] = const_cast<u16 *>(x.an_tt[1]); t.tt2[0] = const_cast<u16 *>(x.an_tt[2]); }
        t.d = d; t.do_norm = do_norm;
        t.tiles0 = (M + 31) / 32; t.tiles01 = t.tiles0 + (B + 31) / 32;
        t.tickets = reinterpret_cast<unsigned *>(ws + p.off_fin); t.n_tickets = FINF_GROUPS + 1;
        const int grid = t.tiles01 + (B + 31) / 32;
        if (d == 32) hipLaunchKernelGGL(prep_tiles_kernel<32>, dim3(grid), dim3(256), 0, st, t);
        else if (d == 64) hipLaunchKernelGGL(prep_tiles_kernel<64>, dim3(grid), dim3(256), 0, st, t);
        else hipLaunchKernelGGL(prep_tiles_kernel<128>, dim3(grid), dim3(256), 0, st, t);
        SSLREC_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(prep_rows3_kernel, dim3(grid_for_rows(M + 2 * B)), dim3(256), 0, st, a);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// forward finish: the split-precision modes (whose preparation launch zeroed the tickets) in ONE launch, exact-fp32 in two
static int finish_fwd(const InfPlan &p, float *ws, const float *zsrc, int n_split, int B, int d, int variant_full, float *loss_out, hipStream_t st) {
    const int variant = variant_full & 0xFF;
    const bool one = inf_precision(variant_full).np != 0;
    hipLaunchKernelGGL(infonce_finish_fwd_kernel, dim3(INF_FIN_BLOCKS), dim3(256), 0, st, ws + p.off_e1s, ws + p.off_e2n, zsrc, n_split, B, d, variant,
                       ws + p.off_z, ws + p.off_part, one ? ws + p.off_fin : (float *)nullptr, loss_out, ws + p.off_misc,
                       inf_precision(variant_full).dyn ? ws + p.off_bias : (const float *)nullptr);
    SSLREC_LAUNCH_CHECK();
    if (one) return 0;
    hipLaunchKernelGGL(infonce_reduce_kernel, dim3(1), dim3(256), 0, st, ws + p.off_part, INF_FIN_BLOCKS, loss_out, ws + p.off_z, B, ws + p.off_misc);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// (the row-major planes of `all` and of the anchors were written by prep_all)
static int run_rowsum(const InfPlan &p, float *ws, int B, int M, int d, int variant, float temp, hipStream_t st) {
    const float *E1s = ws + p.off_e1s, *An = ws + p.off_an;
    float *zpart = ws + p.off_zpart;
    const InfPrec prec = inf_precision(variant);
    if (prec.np == 0)
        return SSLREC_BY_D(launch_rowsum<32>(p, E1s, An, B, M, zpart, st), launch_rowsum<64>(p, E1s, An, B, M, zpart, st),
                           launch_rowsum<128>(p, E1s, An, B, M, zpart, st));
    const X3Planes x = x3_planes(p, ws, B, M, d);
    if (prec.dyn) {
        int rc = SSLREC_BY_D(launch_dyn_bias<32>(p, x, ws, B, M, st), launch_dyn_bias<64>(p, x, ws, B, M, st), launch_dyn_bias<128>(p, x, ws, B, M, st));
        if (rc) return rc;
        const float *ds = ws + p.off_misc + DYN_SC_MUL, *db = ws + p.off_bias;
        return SSLREC_BY_D((launch_rowsum_x3<32, 2, true>(p, x, B, M, zpart, st, temp, ds, db)), (launch_rowsum_x3<64, 2, true>(p, x, B, M, zpart, st, temp, ds, db)),
                           (launch_rowsum_x3<128, 2, true>(p, x, B, M, zpart, st, temp, ds, db)));
    }
    if (prec.f16)
        return SSLREC_BY_D((launch_rowsum_x3<32, 2, true>(p, x, B, M, zpart, st, temp)), (launch_rowsum_x3<64, 2, true>(p, x, B, M, zpart, st, temp)),
                           (launch_rowsum_x3<128, 2, true>(p, x, B, M, zpart, st, temp)));
    if (prec.np == 2)
        return SSLREC_BY_D((launch_rowsum_x3<32, 2>(p, x, B, M, zpart, st)), (launch_rowsum_x3<64, 2>(p, x, B, M, zpart, st)),
                           (launch_rowsum_x3<128, 2>(p, x, B, M, zpart, st)));
    return SSLREC_BY_D((launch_rowsum_x3<32, 3>(p, x, B, M, zpart, st)), (launch_rowsum_x3<64, 3>(p, x, B, M, zpart, st)),
                       (launch_rowsum_x3<128, 3>(p, x, B, M, zpart, st)));
}

// W = sum_j P a_j per column split (the anchor-gradient role; NS = planes of its second product: a cancelling weighted mean -- the
// sensitive one) -- with ZSUM the same launch also leaves the forward pass's partial row sums in zpart.  Needs An and E1s (and, split
// modes, their row-major planes) from prep_all; writes the tile-transposed planes of `all` first (split modes).
template <bool ZSUM>
static int run_anchor_role(const InfPlan &p, float *ws, int B, int M, int d, int variant, float temp, hipStream_t st, bool tt_ready = false) {
    const float *E1s = ws + p.off_e1s, *An = ws + p.off_an;
    float *Wpart = ws + p.off_wpart, *zpart = ws + p.off_zpart;
    const InfPrec prec = inf_precision(variant);
    if (prec.np == 0)
        return SSLREC_BY_D((launch_bwd_anchor<32, ZSUM>(p, E1s, An, B, M, Wpart, zpart, st)), (launch_bwd_anchor<64, ZSUM>(p, E1s, An, B, M, Wpart, zpart, st)),
                           (launch_bwd_anchor<128, false>(p, E1s, An, B, M, Wpart, zpart, st)));      // fwd_w_active(): never ZSUM here
    const X3Planes x = x3_planes(p, ws, B, M, d);      // the row-major planes were written by prep_all
    int rc = tt_ready ? 0 : split_tt(An, M, d, x.an_tt, st, prec.f16, H3_ALL_SCALE,      // (tt_ready: written by the call's preparation launch)
                                     prec.dyn ? ws + p.off_misc + DYN_SC_ALL : (const float *)nullptr);
    if (rc) return rc;
    if (prec.dyn) {
        if (ZSUM) {      // the forward pass: the per-anchor bias is made here; the backward call finds it in the workspace
            rc = SSLREC_BY_D(launch_dyn_bias<32>(p, x, ws, B, M, st), launch_dyn_bias<64>(p, x, ws, B, M, st), launch_dyn_bias<128>(p, x, ws, B, M, st));
            if (rc) return rc;
        }
        const float *dm = ws + p.off_misc, *db = ws + p.off_bias;
        return SSLREC_BY_D((launch_bwd_anchor_x3<32, 2, 2, ZSUM, true>(p, x, B, M, Wpart, zpart, st, temp, dm, db)),
                           (launch_bwd_anchor_x3<64, 2, 2, ZSUM, true>(p, x, B, M, Wpart, zpart, st, temp, dm, db)),
                           (launch_bwd_anchor_x3<128, 2, 2, ZSUM, true>(p, x, B, M, Wpart, zpart, st, temp, dm, db)));
    }
    if (prec.f16)
        return SSLREC_BY_D((launch_bwd_anchor_x3<32, 2, 2, ZSUM, true>(p, x, B, M, Wpart, zpart, st, temp)),
                           (launch_bwd_anchor_x3<64, 2, 2, ZSUM, true>(p, x, B, M, Wpart, zpart, st, temp)),
                           (launch_bwd_anchor_x3<128, 2, 2, ZSUM, true>(p, x, B, M, Wpart, zpart, st, temp)));
#define SSLREC_ANCHOR(NP, NS)                                                                                               \
    SSLREC_BY_D((launch_bwd_anchor_x3<32, NP, NS, ZSUM>(p, x, B, M, Wpart, zpart, st)),                                     \
                (launch_bwd_anchor_x3<64, NP, NS, ZSUM>(p, x, B, M, Wpart, zpart, st)),                                     \
                (launch_bwd_anchor_x3<128, NP, NS, ZSUM>(p, x, B, M, Wpart, zpart, st)))
    if (prec.np == 2 && prec.ns == 2) return SSLREC_ANCHOR(2, 2);
    if (prec.np == 2) return SSLREC_ANCHOR(2, 3);
    if (prec.ns == 2) return SSLREC_ANCHOR(3, 2);
    return SSLREC_ANCHOR(3, 3);
#undef SSLREC_ANCHOR
}

// round 6: does the all-gradient role undo the row normalization in its own epilogue?  (split-precision kernels, normalized variant,
// one anchor split; SSLREC_INFONCE_FOLD=0: the separate pass of rounds 1-5, for A/B measurements)
static bool dall_norm_folded(const InfPlan &p, int variant_full) {
    static const bool on = [] { const char *e = getenv("SSLREC_INFONCE_FOLD"); return !(e && e[0] == '0'); }();
    return on && (variant_full & 0xFF) == 0 && p.n_bsplit == 1 && inf_precision(variant_full).np != 0;
}

// dA = sum_b P V_b (the `all`-gradient role; NS_ALL planes in its second product).  V (fp32 mode: in ws, by make_v; split modes: its
// tile-transposed planes, by make_v_tt) must be ready.
static int run_all_role(const InfPlan &p, float *ws, int B, int M, int d, int variant, float temp, float *dALL, hipStream_t st) {
    const float *E1s = ws + p.off_e1s, *An = ws + p.off_an, *V = ws + p.off_v;
    const InfPrec prec = inf_precision(variant);
    if (prec.np == 0)
        return SSLREC_BY_D(launch_bwd_all<32>(E1s, V, An, B, M, dALL, st), launch_bwd_all<64>(E1s, V, An, B, M, dALL, st),
                           launch_bwd_all<128>(E1s, V, An, B, M, dALL, st));
    const X3Planes x = x3_planes(p, ws, B, M, d);
    float *dst = p.n_bsplit > 1 ? ws + p.off_dapart : dALL;      // (split anchor stream: partials to the slab, summed by finish_dall)
    const int nbs = p.n_bsplit;
    // the normalization's backward in the role's own epilogue (one anchor split, normalized variant): finish_dall then has nothing to do
    const bool fold = dall_norm_folded(p, variant);
    const float *nan_ = fold ? An : nullptr, *nrn = fold ? ws + p.off_rna : nullptr;
    if (prec.dyn) {
        const float *om = ws + p.off_misc + 1, *dm = ws + p.off_misc, *db = ws + p.off_bias;
        return SSLREC_BY_D((launch_bwd_all_x3<32, 2, 2, true>(x, B, M, dst, nbs, st, temp, om, nan_, nrn, dm, db)),
                           (launch_bwd_all_x3<64, 2, 2, true>(x, B, M, dst, nbs, st, temp, om, nan_, nrn, dm, db)),
                           (launch_bwd_all_x3<128, 2, 2, true>(x, B, M, dst, nbs, st, temp, om, nan_, nrn, dm, db)));
    }
    if (prec.f16) {
        const float *om = ws + p.off_misc + 1;
        return SSLREC_BY_D((launch_bwd_all_x3<32, 2, 2, true>(x, B, M, dst, nbs, st, temp, om, nan_, nrn)), (launch_bwd_all_x3<64, 2, 2, true>(x, B, M, dst, nbs, st, temp, om, nan_, nrn)),
                           (launch_bwd_all_x3<128, 2, 2, true>(x, B, M, dst, nbs, st, temp, om, nan_, nrn)));
    }
#define SSLREC_ALL(NP, NS)                                                                                                  \
    SSLREC_BY_D((launch_bwd_all_x3<32, NP, NS>(x, B, M, dst, nbs, st, 1.f, nullptr, nan_, nrn)), (launch_bwd_all_x3<64, NP, NS>(x, B, M, dst, nbs, st, 1.f, nullptr, nan_, nrn)),     \
                (launch_bwd_all_x3<128, NP, NS>(x, B, M, dst, nbs, st, 1.f, nullptr, nan_, nrn)))
    if (prec.np == 2 && prec.ns_all == 2) return SSLREC_ALL(2, 2);
    if (prec.np == 2) return SSLREC_ALL(2, 3);
    if (prec.ns_all == 2) return SSLREC_ALL(3, 2);
    return SSLREC_ALL(3, 3);
#undef SSLREC_ALL
}

// SSLREC_INFONCE_FWD_W is honoured wherever the row sums fit beside the anchor-gradient accumulators; the exact-fp32 kernel at
// d = 128 already uses all 512 registers of a lane (the row sums would spill), so there the flag changes nothing: forward and
// backward both decide with this function
static bool fwd_w_active(int variant_full, int d) {
    return (variant_full & SSLREC_INFONCE_FWD_W) && !(inf_precision(variant_full).np == 0 && d == 128);
}

// after run_all_role: dALL = [row-normalization backward of] the sum of the anchor splits (a launch is needed when there is a
// normalization to undo, variant 0, or a slab to add up)
static int finish_dall(const InfPlan &p, float *ws, int M, int d, int variant_full, float *dALL, hipStream_t st) {
    const int variant = variant_full & 0xFF;
    const bool slab = p.n_bsplit > 1 && inf_precision(variant_full).np != 0;
    if (variant != 0 && !slab) return 0;
    if (dall_norm_folded(p, variant_full)) return 0;      // done in the all-gradient role's epilogue
    hipLaunchKernelGGL(norm_bwd_rows_kernel, dim3(grid_for_rows(M)), dim3(256), 0, st, ws + p.off_an, ws + p.off_rna, M, d, dALL,
                       slab ? ws + p.off_dapart : (const float *)nullptr, p.n_bsplit, variant == 0 ? 1 : 0);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// the forward pass's hot stage: partial row sums -- and, under SSLREC_INFONCE_FWD_W, the anchor-gradient partials with them
static int run_fwd_hot(const InfPlan &p, float *ws, int B, int M, int d, int variant_full, float temp, hipStream_t st) {
    if (fwd_w_active(variant_full, d)) return run_anchor_role<true>(p, ws, B, M, d, variant_full, temp, st, /*tt_ready=*/inf_precision(variant_full).np != 0);
    return run_rowsum(p, ws, B, M, d, variant_full, temp, st);
}

// the backward pass's hot stages: fills Wpart (unless the forward pass already did) and dALL
static int run_bwd_hot(const InfPlan &p, float *ws, int B, int M, int d, int variant_full, float temp, float *dALL, hipStream_t st) {
    if (!fwd_w_active(variant_full, d)) {
        const int rc = run_anchor_role<false>(p, ws, B, M, d, variant_full, temp, st);
        if (rc) return rc;
    }
    return run_all_role(p, ws, B, M, d, variant_full, temp, dALL, st);
}

// backward prologue: V (fp32) or its planes; optionally clears the scatter table of the call in the same launch
static int make_v_any(const InfPlan &p, float *ws, int B, int M, int d, int variant_full, float temp, const float *gscale_dev,
                      const DetTable *tab, hipStream_t st) {
    const int variant = variant_full & 0xFF;
    const float *E1s = ws + p.off_e1s, *Z = ws + p.off_z;
    if (inf_precision(variant_full).np == 0) {
        if (tab) {
            hipLaunchKernelGGL(det_clear_only_kernel, dim3(64), dim3(256), 0, st, *tab);
            SSLREC_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(infonce_make_v_kernel, dim3(grid_for_rows(B)), dim3(256), 0, st, E1s, Z, gscale_dev, B, d, variant, ws + p.off_v);
        SSLREC_LAUNCH_CHECK();
        return 0;
    }
    const X3Planes x = x3_planes(p, ws, B, M, d);
    const size_t total = (size_t)((B + 31) / 32) * 32 * d;
    hipLaunchKernelGGL(make_v_tt_kernel, dim3(grid_for_elems_x3(total)), dim3(256), 0, st, E1s, Z, gscale_dev, B, d, variant,
                       const_cast<u16 *>(x.v_tt[0]), const_cast<u16 *>(x.v_tt[1]), const_cast<u16 *>(x.v_tt[2]), tab ? *tab : DetTable{},
                       tab ? 1 : 0, inf_precision(variant_full).f16 ? 1 : 0, LOG2E_F / temp, h3_bias(temp), ws + p.off_misc,
                       inf_precision(variant_full).dyn ? ws + p.off_bias : (const float *)nullptr);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_infonce_fwd_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                                      int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                                      int32_t variant_full, float *ws, float *loss_out, void *stream) {
    if (temp > 0.f) variant_full = inf_resolve(variant_full, temp);
    if (!inf_args_ok(T1, T2, B, ALL, M, d, temp, variant_full) || !ws || !loss_out) return SSLREC_E_BADARG;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    float *E1s = ws + p.off_e1s, *E2n = ws + p.off_e2n;
    int rc = prep_all(p, ws, T1, i1, T2, i2, B, ALL, M, d, temp, variant_full, st, fwd_w_active(variant_full, d));
    if (rc) return rc;
    rc = run_fwd_hot(p, ws, B, M, d, variant_full, temp, st);
    if (rc) return rc;
    (void)E1s; (void)E2n; (void)variant;
    return finish_fwd(p, ws, ws + p.off_zpart, p.n_split, B, d, variant_full, loss_out, st);
}

extern "C" int sslrec_infonce_bwd_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                                      int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                                      int32_t variant_full, float *ws, const float *gscale_dev, float *dE1,
                                      float *dE2, float *dALL, void *stream) {
    if (temp > 0.f) variant_full = inf_resolve(variant_full, temp);
    if (!inf_args_ok(T1, T2, B, ALL, M, d, temp, variant_full) || !ws || !gscale_dev || !dE1 || !dE2 || !dALL)
        return SSLREC_E_BADARG;
    (void)i1; (void)i2;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    const float *An = ws + p.off_an, *E1s = ws + p.off_e1s, *E2n = ws + p.off_e2n, *Z = ws + p.off_z;
    float *Wpart = ws + p.off_wpart;
    int rc = make_v_any(p, ws, B, M, d, variant_full, temp, gscale_dev, nullptr, st);
    if (rc) return rc;
    rc = run_bwd_hot(p, ws, B, M, d, variant_full, temp, dALL, st);
    if (rc) return rc;
    hipLaunchKernelGGL(infonce_finish_bwd_kernel, dim3(grid_for_rows(B)), dim3(256), 0, st, E1s, E2n,
                       ws + p.off_rn1, ws + p.off_rn2, Wpart, p.n_split, Z, gscale_dev, B, d, temp, variant, dE1,
                       dE2, 0, (const int64_t *)nullptr, (const int64_t *)nullptr, (float *)nullptr, (float *)nullptr, DetTable{});
    SSLREC_LAUNCH_CHECK();
    return finish_dall(p, ws, M, d, variant_full, dALL, st);
}

// backward + the scatter of the gathered rows' gradients in one call (include/sslrec_hip.h): dE [2B, d] receives dE1 then dE2;
// rows of a role with an index array are added into its table (dT1 / dT2, zero-initialised or already holding gradients --
// dT2 may be dALL itself) in a fixed order; the table is cleared by the backward prologue, both roles are registered by one
// launch and reduced by one
extern "C" int sslrec_infonce_bwd_scatter_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2, int32_t B,
                                              const float *ALL, int32_t M, int32_t d, float temp, int32_t variant_full, float *ws,
                                              const float *gscale_dev, float *dE, float *dT1, float *dT2, float *dALL,
                                              void *scatter_ws, void *stream) {
    if (temp > 0.f) variant_full = inf_resolve(variant_full, temp);
    if (!inf_args_ok(T1, T2, B, ALL, M, d, temp, variant_full) || !ws || !gscale_dev || !dE || !dALL || !scatter_ws)
        return SSLREC_E_BADARG;
    if ((size_t)2 * B > DET_MAX || (i1 && !dT1) || (i2 && !dT2)) return SSLREC_E_BADARG;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    const float *An = ws + p.off_an, *E1s = ws + p.off_e1s, *E2n = ws + p.off_e2n, *Z = ws + p.off_z;
    float *Wpart = ws + p.off_wpart;
    const DetTable tab = det_table(scatter_ws);
    int rc = make_v_any(p, ws, B, M, d, variant_full, temp, gscale_dev, &tab, st);
    if (rc) return rc;
    rc = run_bwd_hot(p, ws, B, M, d, variant_full, temp, dALL, st);
    if (rc) return rc;
    float *dE1 = dE, *dE2 = dE + (size_t)B * d;
    const int scatter = (i1 || i2) ? 1 : 0;      // the rows are registered for the deterministic scatter by the same launch
    hipLaunchKernelGGL(infonce_finish_bwd_kernel, dim3(grid_for_rows(B)), dim3(256), 0, st, E1s, E2n,
                       ws + p.off_rn1, ws + p.off_rn2, Wpart, p.n_split, Z, gscale_dev, B, d, temp, variant, dE1, dE2,
                       scatter, i1, i2, dT1, dT2, tab);
    SSLREC_LAUNCH_CHECK();
    rc = finish_dall(p, ws, M, d, variant_full, dALL, st);
    if (rc) return rc;
    if (!scatter) return 0;
    return det_reduce(tab, 2 * B, dE, d, st);
}

// ---------------------------------------------------------------------------------------
// Staged entry points for a ROW-SHARDED `all` table: every rank holds the same B
// anchors/positives and M_local rows of `all`.  Z_b and W_b = sum_j exp(s_bj) a_j are sums over j, so a
// rank produces its partial, the host all-reduces B (resp. B*d) floats, and the finish kernels run on
// the totals.  Same kernels as the single-GPU path; only the split reduction is cut out of the finishers.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_splits_kernel(const float *src, int n_split, size_t n, float *out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < n_split; ++k) s += src[(size_t)k * n + i];
        out[i] = s;
    }
}

static int grid_for_elems(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int sslrec_infonce_shard_rowsum_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                                               int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                                               int32_t variant_full, float *ws, float *z_part, void *stream) {
    if (temp > 0.f) variant_full = inf_resolve(variant_full, temp);
    if (!inf_args_ok(T1, T2, B, ALL, M, d, temp, variant_full) || !ws || !z_part) return SSLREC_E_BADARG;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    int rc = prep_all(p, ws, T1, i1, T2, i2, B, ALL, M, d, temp, variant_full, st, fwd_w_active(variant_full, d));
    if (rc) return rc;
    rc = run_fwd_hot(p, ws, B, M, d, variant_full, temp, st);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_splits_kernel, dim3(grid_for_elems(B)), dim3(256), 0, st, ws + p.off_zpart, p.n_split,
                       (size_t)B, z_part);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_infonce_shard_loss_f32(int32_t B, int32_t M, int32_t d, int32_t variant_full, float *ws,
                                             const float *z_total, float *loss_out, void *stream) {
    if (B <= 0 || M <= 0 || !(d == 32 || d == 64 || d == 128) || !inf_variant_ok(variant_full) || !ws ||
        !z_total || !loss_out)
        return SSLREC_E_BADARG;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    (void)variant;
    return finish_fwd(p, ws, z_total, 1, B, d, variant_full, loss_out, st);
}

extern "C" int sslrec_infonce_shard_bwd_f32(int32_t B, int32_t M, int32_t d, float temp, int32_t variant_full, float *ws,
                                            const float *gscale_dev, float *w_part, float *dALL, void *stream) {
    if (temp > 0.f) variant_full = inf_resolve(variant_full, temp);
    if (B <= 0 || M <= 0 || !(d == 32 || d == 64 || d == 128) || !inf_variant_ok(variant_full) || temp <= 0.f ||
        !ws || !gscale_dev || !w_part || !dALL)
        return SSLREC_E_BADARG;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    const float *An = ws + p.off_an;
    float *Wpart = ws + p.off_wpart;
    int rc = make_v_any(p, ws, B, M, d, variant_full, temp, gscale_dev, nullptr, st);
    if (rc) return rc;
    rc = run_bwd_hot(p, ws, B, M, d, variant_full, temp, dALL, st);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_splits_kernel, dim3(grid_for_elems((size_t)B * d)), dim3(256), 0, st, Wpart, p.n_split,
                       (size_t)B * d, w_part);
    SSLREC_LAUNCH_CHECK();
    return finish_dall(p, ws, M, d, variant_full, dALL, st);
}

extern "C" int sslrec_infonce_shard_finish_bwd_f32(int32_t B, int32_t M, int32_t d, float temp, int32_t variant_full,
                                                   float *ws, const float *gscale_dev, const float *w_total,
                                                   float *dE1, float *dE2, void *stream) {
    if (temp > 0.f) variant_full = inf_resolve(variant_full, temp);
    if (B <= 0 || M <= 0 || !(d == 32 || d == 64 || d == 128) || !inf_variant_ok(variant_full) || temp <= 0.f ||
        !ws || !gscale_dev || !w_total || !dE1 || !dE2)
        return SSLREC_E_BADARG;
    const int variant = variant_full & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    const InfPlan p = make_plan(B, M, d);
    hipLaunchKernelGGL(infonce_finish_bwd_kernel, dim3(grid_for_rows(B)), dim3(256), 0, st, ws + p.off_e1s,
                       ws + p.off_e2n, ws + p.off_rn1, ws + p.off_rn2, w_total, 1, ws + p.off_z, gscale_dev, B, d, temp,
                       variant, dE1, dE2, 0, (const int64_t *)nullptr, (const int64_t *)nullptr, (float *)nullptr, (float *)nullptr, DetTable{});
    SSLREC_LAUNCH_CHECK();
    return 0;
}
