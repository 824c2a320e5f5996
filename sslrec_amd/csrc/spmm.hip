// CSR SpMM for LightGCN-style propagation on MI355X (gfx950).
//
// Replaces torch.spmm(adj, embeds) (reference models/general_cf/lightgcn.py:28-29), its
// autograd backward, LightGCL's gather/index_add_ product (lightgcl.py:58-65), the layer
// SUM (lightgcn.py:41), EmbedPerturb (aug_utils.py:125-132) and EdgeDrop (aug_utils.py:18-31).
//
// Mapping to the hardware
//   * one persistent 64-lane wavefront per work STREAM (~32 streams per CU): the host deals
//     the row segments (a row, or a chunk of a long row) to streams of equal length;
//   * a stream's (col,val) array is wave-uniform, so it is fetched with scalar loads
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
// HBM traffic model: nnz*8 + n_rseg*8 + n_waves*16 + n_cols*d*4 + n_rows*d*4 bytes.
#include "common.h"
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef f32x2 T; };
template <> struct VecT<4> { typedef f32x4 T; };

struct SpmmArgs {
    const int32_t *w_start, *w_len, *r_ptr, *r_len, *r_dst;
    int32_t n_waves;
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

// ---- the stream kernel ------------------------------------------------------------------
// One PERSISTENT wavefront per work stream: the host deals the row segments to n_waves
// (~32 per CU) streams of equal length and lays every stream's entries out contiguously, so a
// wave walks one long (col,val) array with scalar loads, always has U neighbour-row loads in
// flight, and pays the per-row latency chain (metadata -> columns -> rows -> store) once per
// stream instead of once per row.  A scalar counter `rem` tracks the entries left in the
// current row segment; when it reaches 0 the accumulator is written out (fused epilogue) and
// the next segment's (len,dst) is fetched.  Empty segments flush immediately -> exact zeros.
template <int VEC>
__device__ __forceinline__ void emit_row(const SpmmArgs &a, int D, int dst, int off, bool active,
                                         float (&acc)[VEC]) {
    if (dst < 0) {   // chunk of a long row: park the partial sum
        if (active) vec_store<VEC>(a.partial + (size_t)(~dst) * D + off, acc);
    } else {
        finish_row<VEC>(a, D, (size_t)dst, off, active, acc);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
}

#define SSLREC_FLUSH_WHILE_DONE()                              \
    while (rem == 0) {                                         \
        emit_row<VEC>(a, D, dst, off, active, acc);            \
        ++k;                                                   \
        if (k < kend) {                                        \
            rem = a.r_len[k];                                  \
            dst = a.r_dst[k];                                  \
        } else {                                               \
            rem = 0x7fffffff;                                  \
        }                                                      \
    }

// X row of column c for this lane.  BIG=false: the table is < 4 GiB, so a 32-bit byte offset added
// to the (SGPR) table base is enough -- one 32-bit VALU add per edge instead of 64-bit scalar
// shifts/adds on the CU's single scalar unit, which was the measured bottleneck of the first
// version (9 scalar instructions per edge, ~13 clk/edge/CU).
template <int D, int VEC, bool BIG>
__device__ __forceinline__ void load_xrow(float (&dst)[VEC], const float *__restrict__ X, int c, int off) {
    if constexpr (BIG) {
        vec_load<VEC>(dst, X + (size_t)c * D + off);
    } else {
        const uint32_t byte_off = (uint32_t)c * (uint32_t)(D * 4) + (uint32_t)(off * 4);
        vec_load<VEC>(dst, reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + byte_off));
    }
}

template <int D, int U, bool BIG>
__global__ __launch_bounds__(256) void spmm_stream_kernel(SpmmArgs a) {
    constexpr int VEC = (D >= 64) ? D / 64 : 1;
    constexpr int LANES = D / VEC;   // 64 lanes, or 32 at d=32 (upper half mirrors the lower)
    const int lane = threadIdx.x & 63;
    const bool active = lane < LANES;
    const int off = (lane & (LANES - 1)) * VEC;
    const int w = blockIdx.x * 4 + wave_in_block();
    if (w >= a.n_waves) return;
    int e = a.w_start[w];
    const int ee = e + a.w_len[w];
    int k = a.r_ptr[w];
    const int kend = a.r_ptr[w + 1];
    const int32_t *__restrict__ c = a.col;
    const float *__restrict__ v = a.val;
    const float *__restrict__ X = a.X;

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    int rem = 0x7fffffff, dst = 0;
    if (k < kend) {
        rem = a.r_len[k];
        dst = a.r_dst[k];
    }
    SSLREC_FLUSH_WHILE_DONE();   // leading empty segments

    // Software pipeline without register copies: two (col,val) register sets A/B alternate; the
    // set of batch i+1 is requested (scalar loads) while the neighbour rows of batch i are in flight.
    // (The first version copied next->current and zero-initialized the next set every batch:
    // 32 of its ~100 instructions per 8 edges, on a kernel that PMC showed to be issue-bound.)
    int cA[U], cB[U];
    float vA[U], vB[U];
#define SSLREC_STREAM_BATCH(CC, VV, CN, VN)                                                   \
    {                                                                                         \
        float x[U][VEC];                                                                      \
        _Pragma("unroll") for (int j = 0; j < U; ++j) load_xrow<D, VEC, BIG>(x[j], X, CC[j], off); \
        if (e + 2 * U <= ee) {                                                                \
            _Pragma("unroll") for (int j = 0; j < U; ++j) {                                   \
                CN[j] = c[e + U + j];                                                         \
                VN[j] = v[e + U + j];                                                         \
            }                                                                                 \
        }                                                                                     \
        if (rem > U) { /* fast path: no row segment ends inside this batch */                 \
            _Pragma("unroll") for (int j = 0; j < U; ++j)                                     \
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) acc[i] = fmaf(VV[j], x[j][i], acc[i]); \
            rem -= U;                                                                         \
        } else {                                                                              \
            _Pragma("unroll") for (int j = 0; j < U; ++j) {                                   \
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) acc[i] = fmaf(VV[j], x[j][i], acc[i]); \
                --rem;                                                                        \
                SSLREC_FLUSH_WHILE_DONE();                                                    \
            }                                                                                 \
        }                                                                                     \
        e += U;                                                                               \
    }
    if (e + U <= ee) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            cA[j] = c[e + j];
            vA[j] = v[e + j];
        }
    }
    while (true) {
        if (e + U > ee) break;
        SSLREC_STREAM_BATCH(cA, vA, cB, vB)
        if (e + U > ee) break;
        SSLREC_STREAM_BATCH(cB, vB, cA, vA)
    }
#undef SSLREC_STREAM_BATCH
    if (e < ee) {   // wave-uniform tail: < U entries, still issued back to back
        float vt[U];
        float x[U][VEC];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            vt[j] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) x[j][i] = 0.f;
            if (e + j < ee) {
                vt[j] = v[e + j];
                load_xrow<D, VEC, BIG>(x[j], X, c[e + j], off);
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (e + j < ee) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = fmaf(vt[j], x[j][i], acc[i]);
                --rem;
                SSLREC_FLUSH_WHILE_DONE();
            }
        }
    }
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
        int s = s0;
        for (; s + 4 <= s1; s += 4) {   // 4 independent loads in flight, added in slot order
            float p[4][VEC];
#pragma unroll
            for (int j = 0; j < 4; ++j) vec_load<VEC>(p[j], a.partial + (size_t)(s + j) * D + lane * VEC);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += p[j][k];
        }
        for (; s < s1; ++s) {
            float p[VEC];
            vec_load<VEC>(p, a.partial + (size_t)s * D + lane * VEC);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += p[k];
        }
    }
    finish_row<VEC>(a, D, (size_t)row, lane * VEC, active, acc);
}

// ---- edge-drop compaction ---------------------------------------------------------------
// one wavefront per stream; kept entries are packed to the front of the stream, every row
// segment's length becomes its kept count (0 -> the row is written as exact zeros).
__global__ __launch_bounds__(256) void edge_drop_compact_kernel(
    const int32_t *w_start, const int32_t *r_ptr, const int32_t *r_len, int n_waves, const int32_t *col,
    const float *val, const int32_t *edge_map, const uint8_t *keep, float scale, int32_t *col_out,
    float *val_out, int32_t *r_len_out, int32_t *w_len_out) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block();
    if (w >= n_waves) return;
    const int base = w_start[w];
    int in = base;    // wave-uniform read cursor
    int out = base;   // wave-uniform write cursor
    for (int k = r_ptr[w]; k < r_ptr[w + 1]; ++k) {
        const int len = r_len[k];
        int kept = 0;
        for (int e0 = 0; e0 < len; e0 += 64) {
            const int e = e0 + lane;
            bool kp = false;
            int cc = 0;
            float vv = 0.f;
            if (e < len) {
                kp = keep[edge_map[in + e]] != 0;
                cc = col[in + e];
                vv = val[in + e] * scale;
            }
            const unsigned long long m = __ballot(kp);
            if (kp) {
                const int pos = out + kept + __popcll(m & ((1ull << lane) - 1ull));
                col_out[pos] = cc;
                val_out[pos] = vv;
            }
            kept += __popcll(m);
        }
        if (lane == 0) r_len_out[k] = kept;
        in += len;
        out += kept;
    }
    if (lane == 0) w_len_out[w] = out - base;
}

// ---- host launchers -----------------------------------------------------------------------
template <int D, bool BIG>
static int launch_spmm_big(const SpmmArgs &a, const sslrec_csr_t *A, hipStream_t st) {
    const int blocks = (a.n_waves + 3) / 4;
    if (blocks > 0) {
        static const int unroll = [] {   // tuning knob for experiments (default 8)
            const char *e = getenv("SSLREC_SPMM_UNROLL");
            return e ? atoi(e) : 8;
        }();
        if constexpr (D >= 128) {
            hipLaunchKernelGGL((spmm_stream_kernel<D, 4, BIG>), dim3(blocks), dim3(256), 0, st, a);
        } else {
            if (unroll == 4) hipLaunchKernelGGL((spmm_stream_kernel<D, 4, BIG>), dim3(blocks), dim3(256), 0, st, a);
            else if (unroll == 16) hipLaunchKernelGGL((spmm_stream_kernel<D, 16, BIG>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((spmm_stream_kernel<D, 8, BIG>), dim3(blocks), dim3(256), 0, st, a);
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

template <int D>
static int launch_spmm(const SpmmArgs &a, const sslrec_csr_t *A, hipStream_t st) {
    const bool big = (unsigned long long)A->n_cols * (unsigned long long)(D * 4) >= (1ull << 32);
    return big ? launch_spmm_big<D, true>(a, A, st) : launch_spmm_big<D, false>(a, A, st);
}

extern "C" int sslrec_spmm_csr_f32(const sslrec_csr_t *A, const int32_t *col_override,
                                   const float *val_override, const int32_t *r_len_override,
                                   const int32_t *w_len_override, const float *X, int32_t d, float *Y,
                                   const sslrec_epilogue_t *epi, float *partial_ws, void *stream) {
    if (!A || !X) return SSLREC_E_BADARG;
    if (!Y && !(epi && epi->acc_out)) return SSLREC_E_BADARG;
    if (A->n_slots > 0 && !partial_ws) return SSLREC_E_BADARG;
    if (epi && ((epi->acc_in == nullptr) != (epi->acc_out == nullptr))) return SSLREC_E_BADARG;
    if ((r_len_override == nullptr) != (w_len_override == nullptr)) return SSLREC_E_BADARG;
    SpmmArgs a;
    a.w_start = A->w_start;
    a.w_len = w_len_override ? w_len_override : A->w_len;
    a.r_ptr = A->r_ptr;
    a.r_len = r_len_override ? r_len_override : A->r_len;
    a.r_dst = A->r_dst;
    a.n_waves = A->n_waves;
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
                                        float *val_out, int32_t *r_len_out, int32_t *w_len_out,
                                        void *stream) {
    if (!A || !edge_map || !keep || !col_out || !val_out || !r_len_out || !w_len_out) return SSLREC_E_BADARG;
    const int blocks = (A->n_waves + 3) / 4;
    if (blocks > 0) {
        hipLaunchKernelGGL(edge_drop_compact_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           A->w_start, A->r_ptr, A->r_len, A->n_waves, A->col, A->val, edge_map, keep, scale,
                           col_out, val_out, r_len_out, w_len_out);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}
