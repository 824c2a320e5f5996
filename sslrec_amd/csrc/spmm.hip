// CSR SpMM for LightGCN-style propagation on MI355X (gfx950).
//
// Replaces torch.spmm(adj, embeds) (reference models/general_cf/lightgcn.py:28-29), its
// autograd backward, LightGCL's gather/index_add_ product (lightgcl.py:58-65), the layer
// SUM (lightgcn.py:41), EmbedPerturb (aug_utils.py:125-132) and EdgeDrop (aug_utils.py:18-31).
//
// Mapping to the hardware
//   * one 64-lane wavefront per row SEGMENT (a row, or a <= seg_max chunk of a long row);
//   * the segment's (col,val) stream is wave-uniform, so it is fetched with scalar loads
//     into SGPRs and costs no vector issue slots;
//   * the neighbour row X[col,:] is ONE fully coalesced vector load per edge
//     (d=64: 64 lanes x 4 B = 256 B; d=128: 8 B/lane; d=256: 16 B/lane; d=32: two edges
//     per load, one per half-wave), addressed as SGPR base + lane offset;
//   * U independent neighbour loads are kept in flight per wave (latency hiding on top of
//     the up-to-8 waves/SIMD the tiny register footprint allows);
//   * the output row is written once, with the perturbation / layer-sum epilogue fused, so
//     Y and SUM never take an extra pass over HBM;
//   * long rows: partial sums to a scratch slab, combined in slot order by a second
//     kernel -> no atomics, bit-deterministic.
// HBM traffic model: nnz*8 + n_seg*12 + n_cols*d*4 + n_rows*d*4 bytes.
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<4> { typedef f32x4 T; };

struct SpmmArgs {
    const int32_t *seg_dst, *seg_start, *seg_len;
    int32_t n_seg;
    const int32_t *col;
    const float *val;
    const float *X;
    float *Y;
    float *partial;
    const float *noise;
    float eps;
    const float *acc_in;
    float *acc_out;
};

template <int VEC>
__device__ __forceinline__ void vec_load(float (&dst)[VEC], const float *p) {
    typedef typename VecT<VEC>::T V;
    V v = *reinterpret_cast<const V *>(p);
    if constexpr (VEC == 1) {
        dst[0] = v;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = v[i];
    }
}

template <int VEC>
__device__ __forceinline__ void vec_store(float *p, const float (&src)[VEC]) {
    typedef typename VecT<VEC>::T V;
    V v;
    if constexpr (VEC == 1) {
        v = src[0];
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = src[i];
    }
    *reinterpret_cast<V *>(p) = v;
}

// Finish one output row held as VEC floats per lane by LANES lanes (LANES*VEC == D).
// `off` = this lane's float offset inside the row, `active` = lane owns data.
template <int VEC>
__device__ __forceinline__ void finish_row(const SpmmArgs &a, int D, size_t row, int off, bool active,
                                           float (&acc)[VEC]) {
    const size_t base = row * (size_t)D + off;
    if (a.noise) {
        float n[VEC];
        float ss = 0.f;
        if (active) {
            vec_load<VEC>(n, a.noise + base);
#pragma unroll
            for (int i = 0; i < VEC; ++i) ss += n[i] * n[i];
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) n[i] = 0.f;
        }
        ss = wave_sum(ss);
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = acc[i] + ((n[i] / nrm) * sign_f(acc[i])) * a.eps;
    }
    if (!active) return;
    if (a.Y) vec_store<VEC>(a.Y + base, acc);
    if (a.acc_out) {
        float s[VEC];
        vec_load<VEC>(s, a.acc_in + base);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] += acc[i];
        vec_store<VEC>(a.acc_out + base, s);
    }
}

// ---- d >= 64: one edge per vector load -------------------------------------------------
template <int D, int U>
__global__ __launch_bounds__(256) void spmm_seg_kernel(SpmmArgs a) {
    constexpr int VEC = D / 64;
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * 4 + wave_in_block();
    if (seg >= a.n_seg) return;
    const int dst = a.seg_dst[seg];
    const int start = a.seg_start[seg];
    const int len = a.seg_len[seg];
    const int32_t *__restrict__ c = a.col + start;
    const float *__restrict__ v = a.val + start;
    const float *__restrict__ xl = a.X + lane * VEC;

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    // software pipeline: the (col,val) block of iteration i+1 is requested (scalar loads)
    // while the neighbour rows of iteration i are still in flight
    int e = 0;
    int cj[U] = {};
    float vj[U] = {};
    if (U <= len) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            cj[j] = c[j];
            vj[j] = v[j];
        }
    }
    for (; e + U <= len; e += U) {
        float x[U][VEC];
#pragma unroll
        for (int j = 0; j < U; ++j) vec_load<VEC>(x[j], xl + (size_t)cj[j] * D);
        int cn[U] = {};
        float vn[U] = {};
        if (e + 2 * U <= len) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                cn[j] = c[e + U + j];
                vn[j] = v[e + U + j];
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(vj[j], x[j][i], acc[i]);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            cj[j] = cn[j];
            vj[j] = vn[j];
        }
    }
    if (e < len) {   // wave-uniform tail: < U edges, still issued back to back
        float vj[U];
        float x[U][VEC];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            vj[j] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) x[j][i] = 0.f;
            if (e + j < len) {
                vj[j] = v[e + j];
                vec_load<VEC>(x[j], xl + (size_t)c[e + j] * D);
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(vj[j], x[j][i], acc[i]);
    }

    if (dst < 0) {   // chunk of a long row: park the partial sum
        vec_store<VEC>(a.partial + (size_t)(~dst) * D + lane * VEC, acc);
        return;
    }
    finish_row<VEC>(a, D, (size_t)dst, lane * VEC, true, acc);
}

// ---- d == 32: two edges per vector load (one per half-wave) -----------------------------
template <int U>
__global__ __launch_bounds__(256) void spmm_seg_kernel_d32(SpmmArgs a) {
    constexpr int D = 32;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int sl = lane & 31;
    const int seg = blockIdx.x * 4 + wave_in_block();
    if (seg >= a.n_seg) return;
    const int dst = a.seg_dst[seg];
    const int start = a.seg_start[seg];
    const int len = a.seg_len[seg];
    const int32_t *__restrict__ c = a.col + start;
    const float *__restrict__ v = a.val + start;
    const float *__restrict__ xl = a.X + sl;

    float acc = 0.f;
    for (int e = 0; e < len; e += 2 * U) {   // wave-uniform trip count
        float vj[U], x[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e0 = e + 2 * j;
            vj[j] = 0.f;
            x[j] = 0.f;
            if (e0 < len) {   // uniform
                const int c0 = c[e0];
                const float v0 = v[e0];
                const bool has1 = (e0 + 1 < len);
                const int c1 = has1 ? c[e0 + 1] : c0;
                const float v1 = has1 ? v[e0 + 1] : 0.f;
                const int cc = half ? c1 : c0;
                vj[j] = half ? v1 : v0;
                x[j] = xl[(size_t)cc * D];
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc = fmaf(vj[j], x[j], acc);
    }
    acc += __shfl_xor(acc, 32, 64);   // both halves now hold the row

    float r[1] = {acc};
    if (dst < 0) {
        if (half == 0) a.partial[(size_t)(~dst) * D + sl] = acc;
        return;
    }
    finish_row<1>(a, D, (size_t)dst, sl, half == 0, r);
}

// ---- long rows: add the chunk partials in slot order, then the same epilogue ------------
template <int D>
__global__ __launch_bounds__(256) void spmm_long_reduce_kernel(SpmmArgs a, const int32_t *long_row,
                                                               const int32_t *long_ptr, int n_long) {
    constexpr int VEC = (D >= 64) ? D / 64 : 1;
    constexpr int LANES = D / VEC;   // 64, or 32 for d=32
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave_in_block();
    if (i >= n_long) return;
    const int row = long_row[i];
    const int s0 = long_ptr[i], s1 = long_ptr[i + 1];
    const bool active = lane < LANES;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    if (active) {
        for (int s = s0; s < s1; ++s) {
            float p[VEC];
            vec_load<VEC>(p, a.partial + (size_t)s * D + lane * VEC);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += p[k];
        }
    }
    finish_row<VEC>(a, D, (size_t)row, lane * VEC, active, acc);
}

// ---- edge-drop compaction ---------------------------------------------------------------
// one wavefront per segment; kept entries are packed to the front of the segment's slot.
__global__ __launch_bounds__(256) void edge_drop_compact_kernel(
    const int32_t *seg_start, const int32_t *seg_len, int n_seg, const int32_t *col, const float *val,
    const int32_t *edge_map, const uint8_t *keep, float scale, int32_t *col_out, float *val_out,
    int32_t *seg_len_out) {
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * 4 + wave_in_block();
    if (seg >= n_seg) return;
    const int start = seg_start[seg];
    const int len = seg_len[seg];
    int out = 0;   // wave-uniform running count
    for (int e0 = 0; e0 < len; e0 += 64) {
        const int e = e0 + lane;
        bool k = false;
        int cc = 0;
        float vv = 0.f;
        if (e < len) {
            k = keep[edge_map[start + e]] != 0;
            cc = col[start + e];
            vv = val[start + e] * scale;
        }
        const unsigned long long m = __ballot(k);
        if (k) {
            const int pos = out + __popcll(m & ((1ull << lane) - 1ull));
            col_out[start + pos] = cc;
            val_out[start + pos] = vv;
        }
        out += __popcll(m);
    }
    if (lane == 0) seg_len_out[seg] = out;
}

// ---- host launchers -----------------------------------------------------------------------
template <int D>
static int launch_spmm(const SpmmArgs &a, const sslrec_csr_t *A, hipStream_t st) {
    const int blocks = (a.n_seg + 3) / 4;
    if (blocks > 0) {
        if constexpr (D == 32) {
            hipLaunchKernelGGL((spmm_seg_kernel_d32<4>), dim3(blocks), dim3(256), 0, st, a);
        } else if constexpr (D == 64) {
            hipLaunchKernelGGL((spmm_seg_kernel<D, 8>), dim3(blocks), dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((spmm_seg_kernel<D, 4>), dim3(blocks), dim3(256), 0, st, a);
        }
        SSLREC_LAUNCH_CHECK();
    }
    if (A->n_long > 0) {
        hipLaunchKernelGGL((spmm_long_reduce_kernel<D>), dim3((A->n_long + 3) / 4), dim3(256), 0, st, a,
                           A->long_row, A->long_ptr, A->n_long);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sslrec_spmm_csr_f32(const sslrec_csr_t *A, const int32_t *col_override,
                                   const float *val_override, const int32_t *seg_len_override,
                                   const float *X, int32_t d, float *Y, const sslrec_epilogue_t *epi,
                                   float *partial_ws, void *stream) {
    if (!A || !X) return SSLREC_E_BADARG;
    if (!Y && !(epi && epi->acc_out)) return SSLREC_E_BADARG;
    if (A->n_slots > 0 && !partial_ws) return SSLREC_E_BADARG;
    if (epi && ((epi->acc_in == nullptr) != (epi->acc_out == nullptr))) return SSLREC_E_BADARG;
    SpmmArgs a;
    a.seg_dst = A->seg_dst;
    a.seg_start = A->seg_start;
    a.seg_len = seg_len_override ? seg_len_override : A->seg_len;
    a.n_seg = A->n_seg;
    a.col = col_override ? col_override : A->col;
    a.val = val_override ? val_override : A->val;
    a.X = X;
    a.Y = Y;
    a.partial = partial_ws;
    a.noise = epi ? epi->noise : nullptr;
    a.eps = epi ? epi->eps : 0.f;
    a.acc_in = epi ? epi->acc_in : nullptr;
    a.acc_out = epi ? epi->acc_out : nullptr;
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 32: return launch_spmm<32>(a, A, st);
        case 64: return launch_spmm<64>(a, A, st);
        case 128: return launch_spmm<128>(a, A, st);
        case 256: return launch_spmm<256>(a, A, st);
        default: return SSLREC_E_BADARG;
    }
}

extern "C" int sslrec_edge_drop_compact(const sslrec_csr_t *A, const int32_t *edge_map,
                                        const uint8_t *keep, float scale, int32_t *col_out,
                                        float *val_out, int32_t *seg_len_out, void *stream) {
    if (!A || !edge_map || !keep || !col_out || !val_out || !seg_len_out) return SSLREC_E_BADARG;
    const int blocks = (A->n_seg + 3) / 4;
    if (blocks > 0) {
        hipLaunchKernelGGL(edge_drop_compact_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           A->seg_start, A->seg_len, A->n_seg, A->col, A->val, edge_map, keep, scale,
                           col_out, val_out, seg_len_out);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}
