// CSR SpMM for LightGCN-style propagation on MI355X (gfx950).
//
// Replaces torch.spmm(adj, embeds) (reference models/general_cf/lightgcn.py:28-29), its
// autograd backward, LightGCL's gather/index_add_ product (lightgcl.py:58-65), the layer
// SUM (lightgcn.py:41), EmbedPerturb (aug_utils.py:125-132) and EdgeDrop (aug_utils.py:18-31).
//
// Mapping to the hardware (the row-streamed kernel; spmm_swept.hip is the other one)
//   * one persistent 64-lane wavefront per work STREAM (20 per CU, one per resident wavefront): the native plan
//     builder (plan.cpp) deals the row segments (a row, or a chunk of a long row) to streams of equal length;
//   * control is scalar (segment counters in SGPRs), data movement is PACKED: every vector instruction is a
//     16-byte-per-lane load, so one wave instruction fetches G = 256/d neighbour rows (d=64: four 256-byte rows, 16
//     lanes each) and the (col, val) of the next block of 4 loads arrive as one int4 + one float4 per lane;
//   * two row sets and two (col, val) sets alternate so that 8 KiB of neighbour rows are in flight per wave;
//   * the G partial sums of a row are combined with two shuffles at the row end and the output row is written
//     once, with the perturbation (noise read, or computed with Philox: philox.h) / layer-sum epilogue fused, so Y
//     and SUM never take an extra pass over HBM;
//   * long rows: partial sums to a scratch slab, combined in slot order by a second
//     kernel -> no atomics, bit-deterministic.
// HBM traffic model: nnz*8 + n_rseg*8 + n_waves*16 + n_cols*d*4 + n_rows*d*4 bytes.
#include "common.h"
#include "philox.h"
#include "swept_fmt.h"
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
    const uint64_t *philox;      // device-side noise (philox.h) when noise == nullptr
    uint32_t philox_stream;
    unsigned long long *stamp;   // measurement hook (sslrec_debug_stamp_next_launch)
    int32_t prio_mode;           // SSLREC_STREAM_PRIO: 1 = the issue priority of a wave rotates as it advances (see spmm_swept.hip)
    // epilogue extensions (sslrec_epilogue_t): noise of a column slice, regularizer gradient folded into the accumulator
    const float *noise_sumsq;
    int32_t noise_rs, noise_co;  // floats per FULL noise row / this table's column offset in it (0 / 0: the table's own)
    const float *axpy_x;
    float axpy_alpha;
    const float *axpy_scale;
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
    if (a.noise || a.philox) {
        float n[VEC];
        float ss = 0.f;
        if (active) {
            if (a.noise) {
                vec_load<VEC>(n, a.noise + base);
            } else {      // the four uniforms of float group nbase / 4 (index in the FULL noise table); this lane takes its VEC of them
                const size_t nbase = a.noise_rs ? row * (size_t)a.noise_rs + a.noise_co + off : base;
                const float4 u = philox_uniform4(philox_load(a.philox), (uint64_t)(nbase >> 2), a.philox_stream);
                const float u4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int i = 0; i < VEC; ++i) n[i] = u4[(nbase & 3) + i];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) ss += n[i] * n[i];
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) n[i] = 0.f;
        }
        ss = a.noise_sumsq ? a.noise_sumsq[row] : wave_sum(ss);
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
        if (a.axpy_x) {
            const float al = a.axpy_alpha * (a.axpy_scale ? *a.axpy_scale : 1.f);
            float xr[VEC];
            vec_load<VEC>(xr, a.axpy_x + base);
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] = fmaf(al, xr[i], s[i]);
        }
        vec_store<VEC>(a.acc_out + base, s);
    }
}

// ---- the stream kernel ------------------------------------------------------------------
// One PERSISTENT wavefront per work stream (32 per CU).  Control is scalar (one stream per wave,
// counters in SGPRs); DATA movement is packed: every vector instruction is a 16-byte-per-lane
// load, so one wave instruction fetches G = 256/d neighbour rows (d=64: 4 rows of 256 B, each
// read by 16 lanes), and the (col,val) of the next 4 loads arrive as one int4 + one float4 per
// lane.  Measured on MI355X (tools/spmm_sweep.py, X resident in L2): a wave-wide dword load
// costs ~10 clk of the CU's vector-memory pipe whether it carries 128 or 256 bytes, a 16-byte
// one ~31 clk for 1 KiB -- the scalar-col / dword-row version was bound by exactly that.
//
// Layout (built by sslrec_amd/graph.py for one embedding size): a stream is a sequence of
// SLOTS; slot k belongs to load k/G and to lane group ("sub") k%G.  Row segments occupy whole
// loads (padded with col = -1, which a lane skips -> no 0*x term is ever formed); 4 loads form
// a block and the block is stored sub-major ([sub][load]) so that a lane reads its 4 columns
// with one 16-byte load.  The G partial sums of a row are combined with 2 shuffles at row end.
typedef int i32x4 __attribute__((ext_vector_type(4)));

// UNCONDITIONAL load: a pad (col = -1, val = 0) fetches row 0's slice.  A predicated load sits in a branch, and
// behind a load that may or may not have been issued the compiler can only wait with `s_waitcnt vmcnt(0)`, i.e. also for the loads
// issued for the NEXT block -- the loop would not be software-pipelined at all (round 3, found in the column-swept kernel's ISA).
// The consumer multiplies through madd0(): an element whose VALUE is zero -- a pad, or an edge a view dropped by zeroing its value
// (row-bundled layout) -- contributes exactly nothing even when the row it happened to fetch holds Inf / NaN (0 * Inf would put a
// NaN into rows that are not neighbours of the diverged row at all: ADVICE r03).
__device__ __forceinline__ void madd0(f32x4 &acc, const float v, const f32x4 &x) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    acc += v * (v == 0.f ? z : x);
}


template <int D, bool BIG>
__device__ __forceinline__ f32x4 load_xslice(const float *__restrict__ X, int c, int sl) {
    c = c < 0 ? 0 : c;
    if constexpr (BIG) {
        return *reinterpret_cast<const f32x4 *>(X + (size_t)c * D + sl * 4);
    } else {   // table < 4 GiB: 32-bit byte offset on the uniform base, one VALU op
        const uint32_t boff = (uint32_t)c * (uint32_t)(D * 4) + (uint32_t)(sl * 16);
        return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(X) + boff);
    }
}

// finish one row segment: add the G partial sums, run the fused epilogue, store (lanes of sub 0)
template <int D>
__device__ __forceinline__ void emit_row(const SpmmArgs &a, int dst, int sub, int sl, f32x4 &acc) {
    constexpr int LPR = D / 4;   // lanes per row slice
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        acc[0] += __shfl_xor(acc[0], o, 64);
        acc[1] += __shfl_xor(acc[1], o, 64);
        acc[2] += __shfl_xor(acc[2], o, 64);
        acc[3] += __shfl_xor(acc[3], o, 64);
    }
    const bool owner = (sub == 0);
    if (dst < 0) {   // chunk of a long row: park the partial sum
        if (owner) *reinterpret_cast<f32x4 *>(a.partial + (size_t)(~dst) * D + sl * 4) = acc;
    } else {
        const size_t base = (size_t)dst * D + sl * 4;
        if (a.noise || a.philox) {
            f32x4 n = {0.f, 0.f, 0.f, 0.f};
            if (owner) {
                if (a.noise) {
                    n = *reinterpret_cast<const f32x4 *>(a.noise + base);
                } else {
                    const size_t nbase = a.noise_rs ? (size_t)dst * a.noise_rs + a.noise_co + sl * 4 : base;
                    const float4 u = philox_uniform4(philox_load(a.philox), (uint64_t)(nbase >> 2), a.philox_stream);
                    n = f32x4{u.x, u.y, u.z, u.w};
                }
            }
            float ss = n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3];
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (a.noise_sumsq) ss = a.noise_sumsq[dst];
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = acc[i] + ((n[i] / nrm) * sign_f(acc[i])) * a.eps;
        }
        if (owner) {
            if (a.Y) *reinterpret_cast<f32x4 *>(a.Y + base) = acc;
            if (a.acc_out) {
                f32x4 t = *reinterpret_cast<const f32x4 *>(a.acc_in + base);
                t += acc;
                if (a.axpy_x) t += (a.axpy_alpha * (a.axpy_scale ? *a.axpy_scale : 1.f)) * *reinterpret_cast<const f32x4 *>(a.axpy_x + base);
                *reinterpret_cast<f32x4 *>(a.acc_out + base) = t;
            }
        }
    }
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
}

#define SSLREC_FLUSH_WHILE_DONE()                    \
    while (rem == 0) {                               \
        emit_row<D>(a, dst, sub, sl, acc);           \
        ++k;                                         \
        if (k < kend) {                              \
            rem = a.r_len[k];                        \
            dst = a.r_dst[k];                        \
        } else {                                     \
            rem = 0x7fffffff;                        \
        }                                            \
    }

template <int D, bool BIG>
__global__ __launch_bounds__(256) void spmm_stream_kernel(SpmmArgs a) {
    constexpr int LPR = D / 4;      // lanes per row slice: 8, 16, 32, 64
    constexpr int G = 64 / LPR;     // rows per vector load:  8, 4, 2, 1
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    const int w = blockIdx.x * 4 + wave_in_block();
    stamp_begin(a.stamp);
    if (w >= a.n_waves) return;
    const int nload = a.w_len[w];                 // loads in this stream
    const int nblk = (nload + 3) >> 2;            // blocks of 4 loads
    int k = a.r_ptr[w];
    const int kend = a.r_ptr[w + 1];
    // this lane's 4 (col,val) of block b live at element  w_start + b*4G + sub*4
    const i32x4 *__restrict__ cq = reinterpret_cast<const i32x4 *>(a.col + a.w_start[w]) + sub;
    const f32x4 *__restrict__ vq = reinterpret_cast<const f32x4 *>(a.val + a.w_start[w]) + sub;
    const float *__restrict__ X = a.X;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int rem = 0x7fffffff, dst = 0;                // rem counts LOADS left in the current row segment
    if (k < kend) {
        rem = a.r_len[k];
        dst = a.r_dst[k];
    }
    SSLREC_FLUSH_WHILE_DONE();                    // leading empty segments

    // two (col,val) sets and two row sets alternate: while block i is consumed, the rows of block
    // i+1 are in flight (8 x 1 KiB per wave) and the (col,val) of block i+2 are being fetched
    i32x4 cA, cB;
    f32x4 vA, vB, vT;
    f32x4 xA[4], xB[4];
#define SSLREC_ISSUE(XX, CC)                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) XX[j] = load_xslice<D, BIG>(X, CC[j], sl);
#define SSLREC_CONSUME(XX)                                              \
    if (rem > 4) {                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) madd0(acc, vT[j], XX[j]); \
        rem -= 4;                                                       \
    } else {                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                 \
            madd0(acc, vT[j], XX[j]);                                   \
            --rem;                                                      \
            SSLREC_FLUSH_WHILE_DONE();                                  \
        }                                                               \
    }
    // Every load of the loop is UNCONDITIONAL (block indices are clamped: the last iterations re-read the last block and gather rows
    // nobody consumes): behind a load in a branch the compiler can only wait with vmcnt(0), which would also wait for the block
    // issued for the NEXT iteration
    if (nblk > 0) {
        const int last = nblk - 1;
        cA = cq[0];
        vA = vq[0];
        cB = cq[(size_t)min(1, last) * G];
        vB = vq[(size_t)min(1, last) * G];
        SSLREC_ISSUE(xA, cA)
        int i = 0;
        const int wq = (int)(blockIdx.x & 3);      // co-resident waves come from different workgroups: spread the phases over them
        while (true) {
            if (a.prio_mode && (i & 14) == 0) {      // every 16 blocks of 4 loads
                switch (((i >> 4) + wq) & 3) {
                    case 0: __builtin_amdgcn_s_setprio(0); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
            SSLREC_ISSUE(xB, cB)
            vT = vA;
            {
                const size_t n2 = (size_t)min(i + 2, last) * G;
                cA = cq[n2];
                vA = vq[n2];
            }
            SSLREC_CONSUME(xA)
            if (++i >= nblk) break;
            SSLREC_ISSUE(xA, cA)
            vT = vB;
            {
                const size_t n3 = (size_t)min(i + 2, last) * G;
                cB = cq[n3];
                vB = vq[n3];
            }
            SSLREC_CONSUME(xB)
            if (++i >= nblk) break;
        }
    }
#undef SSLREC_ISSUE
#undef SSLREC_CONSUME
    stamp_end<false>(a.stamp);      // (the long-row reduce kernel behind it is not part of the stamp)
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
// one wavefront per stream.  Slot k of a stream lives at element
//     (k / (4G)) * 4G + (k % G) * 4 + (k / G) % 4        (blocks of 4 loads, stored sub-major)
// Kept entries of every row segment are re-packed into consecutive slots, each segment padded
// with col = -1 to whole loads; r_len_out / w_len_out are in LOADS.  A fully dropped row keeps
// a zero-length segment and is written as exact zeros by the SpMM.
__device__ __forceinline__ int slot_elem(int k, int G) {
    const int load = k / G, sub = k - load * G;
    return (load >> 2) * (4 * G) + sub * 4 + (load & 3);
}

__global__ __launch_bounds__(256) void edge_drop_compact_kernel(
    const int32_t *w_start, const int32_t *r_ptr, const int32_t *r_len, int n_waves, int G, const int32_t *col,
    const float *val, const int32_t *edge_map, const uint8_t *keep, float keep_rate, const uint64_t *philox,
    uint32_t philox_stream, float scale, int32_t *col_out, float *val_out, int32_t *r_len_out, int32_t *w_len_out) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block();
    if (w >= n_waves) return;
    PhiloxKey pkey = {};
    if (!keep) pkey = philox_load(philox);      // computed mask: floor(u + keep_rate) != 0 (aug_utils.py:28-29)
    const int base = w_start[w];
    int in_slot = 0;    // wave-uniform cursors, in slots
    int out_slot = 0;
    for (int k = r_ptr[w]; k < r_ptr[w + 1]; ++k) {
        const int n = r_len[k] * G;   // slots of this segment, pads included
        int kept = 0;
        for (int s0 = 0; s0 < n; s0 += 64) {
            const int s = s0 + lane;
            bool kp = false;
            int cc = -1;
            float vv = 0.f;
            if (s < n) {
                const int e = base + slot_elem(in_slot + s, G);
                cc = col[e];
                if (cc >= 0) {
                    const int ke = edge_map[e];
                    kp = keep ? keep[ke] != 0 : floorf(philox_uniform1(pkey, (uint64_t)ke, philox_stream) + keep_rate) != 0.f;
                    vv = val[e] * scale;
                }
            }
            const unsigned long long m = __ballot(kp);
            if (kp) {
                const int o = base + slot_elem(out_slot + kept + __popcll(m & ((1ull << lane) - 1ull)), G);
                col_out[o] = cc;
                val_out[o] = vv;
            }
            kept += __popcll(m);
        }
        const int padded = (kept + G - 1) / G * G;
        if (lane < padded - kept) {   // at most G-1 <= 7 pad slots
            const int o = base + slot_elem(out_slot + kept + lane, G);
            col_out[o] = -1;
            val_out[o] = 0.f;
        }
        if (lane == 0) r_len_out[k] = padded / G;
        in_slot += n;
        out_slot += padded;
    }
    // the SpMM reads whole blocks of 4 loads: blank the rest of the last one
    const int blk_slots = 4 * G;
    const int tail = (out_slot + blk_slots - 1) / blk_slots * blk_slots - out_slot;
    if (lane < tail) {   // tail < 4G <= 32
        const int o = base + slot_elem(out_slot + lane, G);
        col_out[o] = -1;
        val_out[o] = 0.f;
    }
    if (lane == 0) w_len_out[w] = out_slot / G;
}

// ---- host launchers -----------------------------------------------------------------------
template <int D, bool BIG>
static int launch_spmm_big(const SpmmArgs &a, const sslrec_csr_t *A, hipStream_t st) {
    const int blocks = (a.n_waves + 3) / 4;
    if (blocks > 0) {
        hipLaunchKernelGGL((spmm_stream_kernel<D, BIG>), dim3(blocks), dim3(256), 0, st, a);
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
    // 64-bit addressing once the operand reaches 4 GiB (SSLREC_SPMM_FORCE_BIG=1 forces it, for tests)
    static const bool force_big = [] { const char *e = getenv("SSLREC_SPMM_FORCE_BIG"); return e && atoi(e) != 0; }();
    const bool big = force_big || (unsigned long long)A->n_cols * (unsigned long long)(D * 4) >= (1ull << 32);
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
    if (epi && epi->n_sum_in != 0) return SSLREC_E_BADARG;      // the deferred layer sum is the column-swept kernel's
    if (epi && epi->scale_flags != 0) return SSLREC_E_BADARG;   // ... and so is the factorized normalization
    if ((r_len_override == nullptr) != (w_len_override == nullptr)) return SSLREC_E_BADARG;
    if (A->d != d) return SSLREC_E_BADARG;   // the packed layout is specific to one embedding size
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
    a.philox = (epi && !epi->noise) ? epi->philox : nullptr;
    a.philox_stream = epi ? epi->philox_stream : 0;
    a.eps = epi ? epi->eps : 0.f;
    a.acc_in = epi ? epi->acc_in : nullptr;
    a.acc_out = epi ? epi->acc_out : nullptr;
    a.noise_sumsq = nullptr; a.noise_rs = a.noise_co = 0; a.axpy_x = nullptr; a.axpy_alpha = 0.f; a.axpy_scale = nullptr;
    if (epi) {
        if (((epi->noise_row_stride | epi->noise_col_off) & 3) || (epi->noise_row_stride && !epi->noise_sumsq) || (epi->axpy_x && !epi->acc_out))
            return SSLREC_E_BADARG;
        a.noise_sumsq = epi->noise_sumsq; a.noise_rs = epi->noise_row_stride; a.noise_co = epi->noise_col_off;
        a.axpy_x = epi->axpy_x; a.axpy_alpha = epi->axpy_alpha; a.axpy_scale = epi->axpy_scale;
    }
    a.stamp = sslrec_take_stamp();
    static const int prio = [] { const char *e = getenv("SSLREC_STREAM_PRIO"); return e ? atoi(e) : 0; }();
    a.prio_mode = prio;
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 32: return launch_spmm<32>(a, A, st);
        case 64: return launch_spmm<64>(a, A, st);
        case 128: return launch_spmm<128>(a, A, st);
        case 256: return launch_spmm<256>(a, A, st);
        default: return SSLREC_E_BADARG;
    }
}

// ---- narrow tables: the row-bundled kernel (sslrec_bundled_t, include/sslrec_hip.h) ---------------------------------------
// d = 8 / 16 (/ 32): one 16-byte-per-lane wave instruction gathers G = 256/d neighbour rows, so a lane group (LPG = d/4 lanes)
// owns a whole OUTPUT row for the length of a bundle and keeps its sum in registers -- no shuffle reduction at the row end
// (the 8-row reduction of the kernel above would cost 20 shuffles per row at d = 8), no LDS, and control stays wave-uniform:
// the rows of a bundle were sorted by length, the bundle runs for the longest one's steps.  The stream format is the
// column-swept kernel's (swept_fmt.h): one coalesced dword load per array = S steps of all G rows, DPP broadcast per step.
// Two metadata blocks and one block of gathers (S wave instructions = S KiB) are in flight per wave; 8 waves per SIMD.
struct BundleArgs {
    const int32_t *col;
    const float *val;
    const int32_t *w_start, *w_ptr, *b_steps, *b_dst;
    const int32_t *w_nblk;         // a compacted view (sslrec_bundled_compact): 64-dword blocks of stream w that are in use; NULL = all of them
    int32_t n_waves;
    const float *X;
    float *Y;
    float *partial;
    const float *noise;
    float eps;
    const float *acc_in;
    float *acc_out;
    const uint64_t *philox;
    uint32_t philox_stream;
    unsigned long long *stamp;
    int32_t prio_mode;
    const float *noise_sumsq;
    int32_t noise_rs, noise_co;
    const float *axpy_x;
    float axpy_alpha;
    const float *axpy_scale;
};

template <int D>
__device__ __forceinline__ void bundle_emit(const BundleArgs &a, int dst, int sl, f32x4 &acc) {
    constexpr int LPG = D / 4;
    const bool row = dst >= 0;
    const size_t base = (size_t)(row ? dst : 0) * D + sl * 4;
    if (a.noise || a.philox) {      // EmbedPerturb (aug_utils.py:125-132): the norm runs over the row's LPG lanes
        f32x4 n = {0.f, 0.f, 0.f, 0.f};
        if (row) {
            if (a.noise) {
                n = *reinterpret_cast<const f32x4 *>(a.noise + base);
            } else {
                const size_t nbase = a.noise_rs ? (size_t)dst * a.noise_rs + a.noise_co + sl * 4 : base;
                const float4 u = philox_uniform4(philox_load(a.philox), (uint64_t)(nbase >> 2), a.philox_stream);
                n = f32x4{u.x, u.y, u.z, u.w};
            }
        }
        float ss = n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3];
#pragma unroll
        for (int o = LPG / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (a.noise_sumsq && row) ss = a.noise_sumsq[dst];
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        if (row) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = acc[i] + ((n[i] / nrm) * sign_f(acc[i])) * a.eps;
        }
    }
    if (row) {
        if (a.Y) *reinterpret_cast<f32x4 *>(a.Y + base) = acc;
        if (a.acc_out) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(a.acc_in + base);
            t += acc;
            if (a.axpy_x) t += (a.axpy_alpha * (a.axpy_scale ? *a.axpy_scale : 1.f)) * *reinterpret_cast<const f32x4 *>(a.axpy_x + base);
            *reinterpret_cast<f32x4 *>(a.acc_out + base) = t;
        }
    } else if (dst != SSLREC_BUNDLE_NONE) {      // chunk of a long row: park the partial sum
        *reinterpret_cast<f32x4 *>(a.partial + (size_t)(~dst) * D + sl * 4) = acc;
    }
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int D, bool BIG>
__global__ __launch_bounds__(256) void spmm_bundle_kernel(BundleArgs a) {
    constexpr int LPG = D / 4, G = 64 / LPG, S = SweptFmt<D>::S;
    static_assert(S <= 8, "row bundles are for narrow tables");
    const int lane = threadIdx.x & 63;
    const int g = lane / LPG, sl = lane % LPG;
    const int w = blockIdx.x * 4 + wave_in_block();
    stamp_begin(a.stamp);
    if (w >= a.n_waves) return;
    int bi = a.w_ptr[w];
    const int be = a.w_ptr[w + 1];
    const int nblk = a.w_nblk ? a.w_nblk[w] : (a.w_start[w + 1] - a.w_start[w]) / 64;
    const int32_t *__restrict__ pc = a.col + a.w_start[w] + lane;
    const float *__restrict__ pv = a.val + a.w_start[w] + lane;
    const float *__restrict__ X = a.X;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int rem = 0x7fffffff, dst = SSLREC_BUNDLE_NONE, dst_n = SSLREC_BUNDLE_NONE;      // rem: blocks left in the current bundle
    if (bi < be) {
        rem = a.b_steps[bi] / S;
        dst = a.b_dst[(size_t)bi * G + g];
        if (bi + 1 < be) dst_n = a.b_dst[(size_t)(bi + 1) * G + g];      // the next bundle's rows arrive while this one runs
    }
#define BD_ADVANCE()                                                                        \
    while (rem == 0) {                                                                      \
        bundle_emit<D>(a, dst, sl, acc);                                                    \
        ++bi;                                                                               \
        if (bi >= be) { rem = 0x7fffffff; break; }                                          \
        rem = a.b_steps[bi] / S;                                                            \
        dst = dst_n;                                                                        \
        dst_n = (bi + 1 < be) ? a.b_dst[(size_t)(bi + 1) * G + g] : SSLREC_BUNDLE_NONE;     \
    }
    BD_ADVANCE()                                   // leading bundles without entries: rows of zeros
    int cA = -1, cB = -1;
    float vA = 0.f, vB = 0.f, vT;
    f32x4 xA[S], xB[S];
#define BD_ISSUE1(XX, CC, J) if constexpr (S > J) XX[J] = load_xslice<D, BIG>(X, sw_bcast<D, (J < S ? J : 0)>(CC), sl);
#define BD_ISSUE(XX, CC) BD_ISSUE1(XX, CC, 0) BD_ISSUE1(XX, CC, 1) BD_ISSUE1(XX, CC, 2) BD_ISSUE1(XX, CC, 3) \
                         BD_ISSUE1(XX, CC, 4) BD_ISSUE1(XX, CC, 5) BD_ISSUE1(XX, CC, 6) BD_ISSUE1(XX, CC, 7)
#define BD_FMA1(XX, J) if constexpr (S > J) madd0(acc, __int_as_float(sw_bcast<D, (J < S ? J : 0)>(__float_as_int(vT))), XX[J]);
#define BD_CONSUME(XX) BD_FMA1(XX, 0) BD_FMA1(XX, 1) BD_FMA1(XX, 2) BD_FMA1(XX, 3) BD_FMA1(XX, 4) BD_FMA1(XX, 5) BD_FMA1(XX, 6) BD_FMA1(XX, 7) \
                       --rem;                                                                                                       \
                       BD_ADVANCE()
    // (unconditional loads with clamped block indices, like the stream kernel above)
    if (nblk > 0) {
        const int last = nblk - 1;
        cA = pc[0]; vA = pv[0];
        cB = pc[(size_t)min(1, last) * 64]; vB = pv[(size_t)min(1, last) * 64];
        BD_ISSUE(xA, cA)
        const int wq = (int)(blockIdx.x & 3);
        int k = 0;
        while (true) {
            if (a.prio_mode && (k & 62) == 0) {      // every 64 blocks
                switch (((k >> 6) + wq) & 3) {
                    case 0: __builtin_amdgcn_s_setprio(0); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
            BD_ISSUE(xB, cB)
            vT = vA;
            { const size_t n2 = (size_t)min(k + 2, last) * 64; cA = pc[n2]; vA = pv[n2]; }
            BD_CONSUME(xA)
            if (++k >= nblk) break;
            BD_ISSUE(xA, cA)
            vT = vB;
            { const size_t n3 = (size_t)min(k + 2, last) * 64; cB = pc[n3]; vB = pv[n3]; }
            BD_CONSUME(xB)
            if (++k >= nblk) break;
        }
    }
#undef BD_ISSUE1
#undef BD_ISSUE
#undef BD_FMA1
#undef BD_CONSUME
#undef BD_ADVANCE
    stamp_end<false>(a.stamp);
}

template <int D>
static int launch_bundled(const BundleArgs &a, const sslrec_bundled_t *A, hipStream_t st) {
    static const bool force_big = [] { const char *e = getenv("SSLREC_SPMM_FORCE_BIG"); return e && atoi(e) != 0; }();
    const bool big = force_big || (unsigned long long)A->n_cols * (unsigned long long)(D * 4) >= (1ull << 32);
    const int blocks = (a.n_waves + 3) / 4;
    if (blocks > 0) {
        if (big) hipLaunchKernelGGL((spmm_bundle_kernel<D, true>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((spmm_bundle_kernel<D, false>), dim3(blocks), dim3(256), 0, st, a);
        SSLREC_LAUNCH_CHECK();
    }
    if (A->n_long > 0) {      // chunk partials in slot order + the same epilogue (the long-row kernel of the streamed layout)
        SpmmArgs r = {};
        r.Y = a.Y; r.partial = a.partial; r.noise = a.noise; r.eps = a.eps; r.acc_in = a.acc_in; r.acc_out = a.acc_out;
        r.philox = a.philox; r.philox_stream = a.philox_stream;
        r.noise_sumsq = a.noise_sumsq; r.noise_rs = a.noise_rs; r.noise_co = a.noise_co;
        r.axpy_x = a.axpy_x; r.axpy_alpha = a.axpy_alpha; r.axpy_scale = a.axpy_scale;
        hipLaunchKernelGGL((spmm_long_reduce_kernel<D>), dim3((A->n_long + 3) / 4), dim3(256), 0, st, r, A->long_row, A->long_ptr,
                           A->n_long);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sslrec_spmm_bundled_f32(const sslrec_bundled_t *A, const float *val_override, const float *X, int32_t d, float *Y,
                                       const sslrec_epilogue_t *epi, float *partial_ws, void *stream) {
    return sslrec_spmm_bundled_view_f32(A, nullptr, val_override, nullptr, nullptr, X, d, Y, epi, partial_ws, stream);
}

extern "C" int sslrec_spmm_bundled_view_f32(const sslrec_bundled_t *A, const int32_t *col_override, const float *val_override,
                                            const int32_t *b_steps_override, const int32_t *w_blocks_override, const float *X, int32_t d,
                                            float *Y, const sslrec_epilogue_t *epi, float *partial_ws, void *stream) {
    if (!A || !X || A->d != d) return SSLREC_E_BADARG;
    if ((b_steps_override == nullptr) != (w_blocks_override == nullptr) || (b_steps_override && !col_override)) return SSLREC_E_BADARG;
    if (!Y && !(epi && epi->acc_out)) return SSLREC_E_BADARG;
    if (A->n_slots > 0 && !partial_ws) return SSLREC_E_BADARG;
    if (epi && ((epi->acc_in == nullptr) != (epi->acc_out == nullptr))) return SSLREC_E_BADARG;
    if (epi && epi->n_sum_in != 0) return SSLREC_E_BADARG;      // the deferred layer sum is the column-swept kernel's
    if (epi && epi->scale_flags != 0) return SSLREC_E_BADARG;   // ... and so is the factorized normalization
    BundleArgs a = {};
    a.col = col_override ? col_override : A->col;
    a.val = val_override ? val_override : A->val;
    a.w_start = A->w_start; a.w_ptr = A->w_ptr; a.b_steps = b_steps_override ? b_steps_override : A->b_steps; a.b_dst = A->b_dst;
    a.w_nblk = w_blocks_override;
    a.n_waves = A->n_waves;
    a.X = X; a.Y = Y; a.partial = partial_ws;
    a.noise = epi ? epi->noise : nullptr;
    a.philox = (epi && !epi->noise) ? epi->philox : nullptr;
    a.philox_stream = epi ? epi->philox_stream : 0;
    a.eps = epi ? epi->eps : 0.f;
    a.acc_in = epi ? epi->acc_in : nullptr;
    a.acc_out = epi ? epi->acc_out : nullptr;
    if (epi) {
        if (((epi->noise_row_stride | epi->noise_col_off) & 3) || (epi->noise_row_stride && !epi->noise_sumsq) || (epi->axpy_x && !epi->acc_out))
            return SSLREC_E_BADARG;
        a.noise_sumsq = epi->noise_sumsq; a.noise_rs = epi->noise_row_stride; a.noise_co = epi->noise_col_off;
        a.axpy_x = epi->axpy_x; a.axpy_alpha = epi->axpy_alpha; a.axpy_scale = epi->axpy_scale;
    }
    a.stamp = sslrec_take_stamp();
    static const int prio = [] { const char *e = getenv("SSLREC_STREAM_PRIO"); return e ? atoi(e) : 0; }();
    a.prio_mode = prio;
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 8: return launch_bundled<8>(a, A, st);
        case 16: return launch_bundled<16>(a, A, st);
        case 32: return launch_bundled<32>(a, A, st);
        default: return SSLREC_E_BADARG;
    }
}

static int edge_drop_compact_any(const sslrec_csr_t *A, const int32_t *edge_map, const uint8_t *keep, float keep_rate,
                                 const uint64_t *philox, uint32_t philox_stream, float scale, int32_t *col_out,
                                 float *val_out, int32_t *r_len_out, int32_t *w_len_out, void *stream) {
    if (!A || !edge_map || (!keep && !philox) || !col_out || !val_out || !r_len_out || !w_len_out) return SSLREC_E_BADARG;
    if (A->d != 32 && A->d != 64 && A->d != 128 && A->d != 256) return SSLREC_E_BADARG;
    const int blocks = (A->n_waves + 3) / 4;
    if (blocks > 0) {
        hipLaunchKernelGGL(edge_drop_compact_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           A->w_start, A->r_ptr, A->r_len, A->n_waves, 256 / A->d, A->col, A->val, edge_map, keep,
                           keep_rate, philox, philox_stream, scale, col_out, val_out, r_len_out, w_len_out);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sslrec_edge_drop_compact(const sslrec_csr_t *A, const int32_t *edge_map,
                                        const uint8_t *keep, float scale, int32_t *col_out,
                                        float *val_out, int32_t *r_len_out, int32_t *w_len_out,
                                        void *stream) {
    if (!keep) return SSLREC_E_BADARG;
    return edge_drop_compact_any(A, edge_map, keep, 0.f, nullptr, 0, scale, col_out, val_out, r_len_out, w_len_out, stream);
}

extern "C" int sslrec_edge_drop_compact_philox(const sslrec_csr_t *A, const int32_t *edge_map, float keep_rate,
                                               const uint64_t *philox_state, uint32_t philox_stream, float scale,
                                               int32_t *col_out, float *val_out, int32_t *r_len_out, int32_t *w_len_out,
                                               void *stream) {
    if (!philox_state || !(keep_rate >= 0.f && keep_rate <= 1.f)) return SSLREC_E_BADARG;
    return edge_drop_compact_any(A, edge_map, nullptr, keep_rate, philox_state, philox_stream, scale, col_out, val_out,
                                 r_len_out, w_len_out, stream);
}

// EdgeDrop on the ROW-BUNDLED layout (replaces EdgeDrop.forward, models/aug_utils.py:18-31, for narrow tables beyond the column-swept
// layout: feature-sliced 8 / 16-column tables of more than ~645 k rows): the dropped entries keep their place in the streams and get
// the VALUE zero -- the bundle kernel's madd0() makes a zero-valued element contribute exactly nothing, so the product is the one over
// the kept entries, with the summation order of the full matrix.  (No compaction: a view costs the full stream length.  The bundled
// streams are sized per bundle of G rows of similar length, which a per-row compaction would break up again.)
__global__ __launch_bounds__(256) void bundled_drop_values_kernel(const float *__restrict__ val, const int32_t *__restrict__ edge_map, size_t n_elem,
                                                                 const uint8_t *__restrict__ keep, float keep_rate,
                                                                 const uint64_t *__restrict__ philox, uint32_t philox_stream, float scale,
                                                                 float *__restrict__ val_out) {
    PhiloxKey pkey = {};
    if (!keep) pkey = philox_load(philox);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_elem; e += (size_t)gridDim.x * 256) {
        const int k = edge_map[e];
        float v = 0.f;
        if (k >= 0) {
            const bool kp = keep ? keep[k] != 0 : floorf(philox_uniform1(pkey, (uint64_t)k, philox_stream) + keep_rate) != 0.f;
            if (kp) v = val[e] * scale;
        }
        val_out[e] = v;
    }
}

extern "C" int sslrec_bundled_drop_values(const sslrec_bundled_t *A, const int32_t *edge_map, const uint8_t *keep, float keep_rate,
                                          const uint64_t *philox_state, uint32_t philox_stream, float scale, float *val_out, void *stream) {
    if (!A || !edge_map || (!keep && !philox_state) || !val_out || A->n_elem < 0) return SSLREC_E_BADARG;
    if (!keep && !(keep_rate >= 0.f && keep_rate <= 1.f)) return SSLREC_E_BADARG;
    if (A->n_elem == 0) return 0;
    const size_t n = (size_t)A->n_elem;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(bundled_drop_values_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A->val, edge_map, n, keep, keep_rate,
                       philox_state, philox_stream, scale, val_out);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// The same view COMPACTED (round 5): every row of a bundle keeps only its kept entries -- in their order, so the sums have the bits of
// the zero-valued form -- the bundle runs for the longest of its G compacted rows (rounded to S steps) and the bundles of a stream move
// up behind each other: a keep-0.5 view walks about half of the stream instead of all of it.  One wave per stream; a bundle's rows are
// the wave's G lane groups (lane (g, j) holds step j of a block for row g), a ballot per block ranks the kept entries of a group.
template <int D>
__global__ __launch_bounds__(256) void bundled_compact_kernel(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                             const int32_t *__restrict__ w_start, const int32_t *__restrict__ w_ptr,
                                                             const int32_t *__restrict__ b_steps, int n_waves,
                                                             const int32_t *__restrict__ edge_map, const uint8_t *__restrict__ keep,
                                                             float keep_rate, const uint64_t *__restrict__ philox, uint32_t philox_stream,
                                                             float scale, int32_t *__restrict__ col_out, float *__restrict__ val_out,
                                                             int32_t *__restrict__ b_steps_out, int32_t *__restrict__ w_blocks_out) {
    constexpr int LPG = D / 4, S = SweptFmt<D>::S;
    static_assert(S == LPG, "a lane group holds one block's steps of its row");
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block();
    if (w >= n_waves) return;
    const int g = lane / LPG, j = lane % LPG;
    const unsigned long long gmask = (((1ull << LPG) - 1ull) << (g * LPG));
    PhiloxKey pkey = {};
    if (!keep) pkey = philox_load(philox);
    const size_t base = (size_t)w_start[w];
    size_t in_blk = 0, out_blk = 0;
    for (int bi = w_ptr[w]; bi < w_ptr[w + 1]; ++bi) {
        const int nb = b_steps[bi] / S;
        int count = 0;                                        // kept entries of this lane group's row so far
        for (int k = 0; k < nb; ++k) {
            const size_t e = base + (in_blk + k) * 64 + lane;
            const int c = col[e];
            bool kp = false;
            float v = 0.f;
            if (c >= 0) {
                const int id = edge_map[e];
                kp = keep ? keep[id] != 0 : floorf(philox_uniform1(pkey, (uint64_t)id, philox_stream) + keep_rate) != 0.f;
                v = val[e] * scale;
            }
            const unsigned long long m = __ballot(kp) & gmask;
            if (kp) {
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                const size_t o = base + (out_blk + pos / S) * 64 + g * LPG + pos % S;
                col_out[o] = c;
                val_out[o] = v;
            }
            count += __popcll(m);
        }
        int longest = count;
#pragma unroll
        for (int o = LPG; o < 64; o <<= 1) longest = max(longest, __shfl_xor(longest, o, 64));
        const int steps_out = (longest + S - 1) / S * S;
        for (int pos = count + j; pos < steps_out; pos += LPG) {      // the shorter rows of the bundle: pads
            const size_t o = base + (out_blk + pos / S) * 64 + g * LPG + pos % S;
            col_out[o] = -1;
            val_out[o] = 0.f;
        }
        if (lane == 0) b_steps_out[bi] = steps_out;
        in_blk += nb;
        out_blk += steps_out / S;
    }
    if (lane == 0) w_blocks_out[w] = (int32_t)out_blk;
}

static int bundled_compact_any(const sslrec_bundled_t *A, const int32_t *edge_map, const uint8_t *keep, float keep_rate,
                               const uint64_t *philox_state, uint32_t philox_stream, float scale, int32_t *col_out, float *val_out,
                               int32_t *b_steps_out, int32_t *w_blocks_out, void *stream) {
    if (!A || !edge_map || (!keep && !philox_state) || !col_out || !val_out || !b_steps_out || !w_blocks_out || A->n_elem < 0)
        return SSLREC_E_BADARG;
    if (!keep && !(keep_rate >= 0.f && keep_rate <= 1.f)) return SSLREC_E_BADARG;
    if (A->n_waves <= 0) return 0;
    const int blocks = (A->n_waves + 3) / 4;
#define BD_COMPACT(DD)                                                                                                          \
    hipLaunchKernelGGL(bundled_compact_kernel<DD>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A->col, A->val, A->w_start, A->w_ptr, \
                       A->b_steps, A->n_waves, edge_map, keep, keep_rate, philox_state, philox_stream, scale, col_out, val_out,   \
                       b_steps_out, w_blocks_out)
    switch (A->d) {
        case 8: BD_COMPACT(8); break;
        case 16: BD_COMPACT(16); break;
        case 32: BD_COMPACT(32); break;
        default: return SSLREC_E_BADARG;
    }
#undef BD_COMPACT
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_bundled_compact(const sslrec_bundled_t *A, const int32_t *edge_map, const uint8_t *keep, float scale, int32_t *col_out,
                                      float *val_out, int32_t *b_steps_out, int32_t *w_blocks_out, void *stream) {
    if (!keep) return SSLREC_E_BADARG;
    return bundled_compact_any(A, edge_map, keep, 0.f, nullptr, 0, scale, col_out, val_out, b_steps_out, w_blocks_out, stream);
}

extern "C" int sslrec_bundled_compact_philox(const sslrec_bundled_t *A, const int32_t *edge_map, float keep_rate, const uint64_t *philox_state,
                                             uint32_t philox_stream, float scale, int32_t *col_out, float *val_out, int32_t *b_steps_out,
                                             int32_t *w_blocks_out, void *stream) {
    if (!philox_state) return SSLREC_E_BADARG;
    return bundled_compact_any(A, edge_map, nullptr, keep_rate, philox_state, philox_stream, scale, col_out, val_out, b_steps_out,
                               w_blocks_out, stream);
}
