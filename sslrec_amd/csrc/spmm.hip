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
// Persistent, lane-group-parallel walk of the streamed CSR.
//   * a lane GROUP of d/4 lanes (16 B per lane) owns one stream; a wave carries 256/d groups
//     (4 at d=64), so one vector instruction advances 4 streams: the first version kept the
//     per-edge bookkeeping in scalar registers and was bound by the CU's instruction issue
//     (PMC: 12.6 instructions per edge, 57 % of wave cycles waiting) -- here it is ~3;
//   * the next 8 (col,val) pairs are fetched by 8 distinct lanes of the group with one
//     coalesced dword load each and broadcast inside the group with ds_swizzle;
//   * 8 neighbour-row reads (one 16-byte load per lane, 4*d contiguous bytes per group) are
//     in flight per group before the first is consumed;
//   * a per-lane counter `rem` tracks the entries left in the current row segment; when it
//     hits 0 the accumulator is written out (fused epilogue) under the group's exec mask.
//     The next segment's (len,dst) and, for the fused layer sum, its acc_in row were requested
//     one segment earlier, so a row end costs no memory round trip.
struct GroupRow {
    int rem, dst;      // current row segment: entries left, destination
    int nrem, ndst;    // next segment (prefetched)
    float accin[4];    // acc_in row of the current segment (prefetched when it became current)
};

template <int D, int LPG>
__device__ __forceinline__ void group_emit(const SpmmArgs &a, GroupRow &r, int &k, int kend, int sl,
                                           float (&acc)[4]) {
    const int dst = r.dst;
    if (dst < 0) {   // chunk of a long row: park the partial sum
        vec_store<4>(a.partial + (size_t)(~dst) * D + sl * 4, acc);
    } else {
        const size_t base = (size_t)dst * D + sl * 4;
        if (a.noise) {
            float n[4];
            vec_load<4>(n, a.noise + base);
            float ss = n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3];
#pragma unroll
            for (int o = LPG / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = acc[i] + ((n[i] / nrm) * sign_f(acc[i])) * a.eps;
        }
        if (a.Y) vec_store<4>(a.Y + base, acc);
        if (a.acc_out) {
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = r.accin[i] + acc[i];
            vec_store<4>(a.acc_out + base, t);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = 0.f;
    // advance: the prefetched segment becomes current; request the one after it
    ++k;
    r.rem = r.nrem;
    r.dst = r.ndst;
    if (k < kend && r.dst >= 0 && a.acc_in) vec_load<4>(r.accin, a.acc_in + (size_t)r.dst * D + sl * 4);
    if (k + 1 < kend) {
        r.nrem = a.r_len[k + 1];
        r.ndst = a.r_dst[k + 1];
    } else {
        r.nrem = 0x7fffffff;
    }
}

template <int D, bool BIG>
__device__ __forceinline__ const f32x4 *xrow_ptr(const float *__restrict__ X, int c, int sl) {
    if constexpr (BIG) {
        return reinterpret_cast<const f32x4 *>(X + (size_t)c * D + sl * 4);
    } else {   // table < 4 GiB: 32-bit byte offset on the uniform base
        const uint32_t boff = (uint32_t)c * (uint32_t)(D * 4) + (uint32_t)(sl * 16);
        return reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(X) + boff);
    }
}

template <int D, bool BIG>
__device__ __forceinline__ void stream_batch(const SpmmArgs &a, GroupRow &r, int &k, int kend, int e, int end,
                                             int sl, float (&acc)[4]) {
    constexpr int LPG = D / 4;
    constexpr int AND = (~((LPG < 32 ? LPG : 32) - 1)) & 0x1f;   // ds_swizzle bit-mask mode: keep group bits
    const int nvalid = end - e;                                   // uniform inside a group, may be <= 0
    // loads are unconditional (index clamped into the stream, so the address is always valid);
    // only the CONSUMPTION below is predicated, so no 0*x term of a padded entry is ever added
    int idx = e + (sl & 7);
    idx = idx < end ? idx : end - 1;
    idx = idx > 0 ? idx : 0;
    const int cc = a.col[idx];
    const float vv = a.val[idx];
    f32x4 x[8];
    float vj[8];
#define SSLREC_FETCH(J)                                                                  \
    {                                                                                    \
        const int cj = __builtin_amdgcn_ds_swizzle(cc, AND | ((J) << 5));                \
        vj[J] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(vv), AND | ((J) << 5))); \
        x[J] = *xrow_ptr<D, BIG>(a.X, cj, sl);                                           \
    }
    SSLREC_FETCH(0) SSLREC_FETCH(1) SSLREC_FETCH(2) SSLREC_FETCH(3)
    SSLREC_FETCH(4) SSLREC_FETCH(5) SSLREC_FETCH(6) SSLREC_FETCH(7)
#undef SSLREC_FETCH
    if (__all(nvalid >= 8 && r.rem > 8)) {   // full batch everywhere and no row segment ends inside it
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(vj[j], x[j][i], acc[i]);
        r.rem -= 8;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nvalid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(vj[j], x[j][i], acc[i]);
                --r.rem;
                while (r.rem == 0) group_emit<D, LPG>(a, r, k, kend, sl, acc);
            }
        }
    }
}

template <int D, bool BIG>
__global__ __launch_bounds__(256) void spmm_stream_kernel(SpmmArgs a) {
    constexpr int LPG = D / 4;     // lanes per group: 8, 16, 32, 64
    constexpr int GPW = 64 / LPG;  // groups (streams) per wave
    const int lane = threadIdx.x & 63;
    const int g = lane / LPG, sl = lane % LPG;
    const int sid = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * GPW + g;   // stream of this lane group
    int e = 0, end = 0, k = 0, kend = 0;
    if (sid < a.n_waves) {
        e = a.w_start[sid];
        end = e + a.w_len[sid];
        k = a.r_ptr[sid];
        kend = a.r_ptr[sid + 1];
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    GroupRow r;
    r.rem = r.nrem = 0x7fffffff;
    r.dst = r.ndst = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.accin[i] = 0.f;
    if (k < kend) {
        r.rem = a.r_len[k];
        r.dst = a.r_dst[k];
        if (r.dst >= 0 && a.acc_in) vec_load<4>(r.accin, a.acc_in + (size_t)r.dst * D + sl * 4);
    }
    if (k + 1 < kend) {
        r.nrem = a.r_len[k + 1];
        r.ndst = a.r_dst[k + 1];
    }
    while (r.rem == 0) group_emit<D, LPG>(a, r, k, kend, sl, acc);   // leading empty segments

    while (__any(e < end)) {
        stream_batch<D, BIG>(a, r, k, kend, e, end, sl, acc);
        e += 8;
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

// ---- the sweep kernel ---------------------------------------------------------------------
// Locality-first variant for tables that exceed the 4 MiB per-XCD L2 (amazon-book: X = 37 MB).
// PMC profiling of the stream kernel showed 59 % L2 misses and ~840 MB of fabric reads per
// SpMM against 113 MB of compulsory traffic: every wave walks its rows column-by-column on
// its own schedule, so the whole table is live in every L2 all the time.  Here
//   * ALL row accumulators of the matrix live on chip: one 1024-thread workgroup per CU owns
//     640 rows (d=64) in its 160 KiB of LDS, 10 rows per 16-lane group (16 B per lane);
//   * every group walks ONE stream that holds the entries of its rows sorted by
//     (column block, row, column): all ~16k groups of the chip start at column block 0 and
//     advance one 8-entry batch per iteration, so at any moment the chip is gathering from a
//     narrow window of X (a ~2 MB column block) that stays resident in every XCD's L2 --
//     each X row is fetched from the fabric about once per XCD instead of once per edge;
//   * control is per lane group (vector), 4 entries per wave instruction at d=64: the stream
//     kernel's scalar per-edge bookkeeping saturated the CU's instruction issue;
//   * (col,val) are fetched 8 at a time by distinct lanes and broadcast inside the group with
//     ds_swizzle; the products are added to the LDS accumulators by ds_read_b128 / 4 fma /
//     ds_write_b128 (in-order per wave, one owner per row -> deterministic);
//   * rows are written once at the end with the fused epilogue, long rows via the partial slab.
struct SweepArgs {
    const int32_t *s_start, *s_len;   // [n_groups]
    const int32_t *cs;                // [nnz] column | slot << 27
    const float *val;
    const int32_t *g_dst;             // [n_groups * SWEEP_SPG]
    int32_t n_groups;
};

#define SWEEP_WPW 16          // waves per workgroup (1024 threads, one workgroup per CU)
#define SWEEP_SPG 10          // accumulator rows per lane group
#define SWEEP_COL_BITS 27
#define SWEEP_UNUSED ((int32_t)0x80000000)

template <int PATTERN>
__device__ __forceinline__ int swz_i(int v) { return __builtin_amdgcn_ds_swizzle(v, PATTERN); }
template <int PATTERN>
__device__ __forceinline__ float swz_f(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN));
}

template <int LPG>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int D, bool PRED>
__device__ __forceinline__ void sweep_batch(const SweepArgs &a, const char *xb, float *my, int e, int nvalid,
                                            int sl, const int32_t *__restrict__ cs_arr,
                                            const float *__restrict__ val_arr) {
    constexpr int LPG = D / 4;
    constexpr int AND = (~((LPG < 32 ? LPG : 32) - 1)) & 0x1f;   // ds_swizzle bit-mask mode: keep group bits
    const int q = sl & 7;
    int cs = 0;
    float vv = 0.f;
    if (!PRED || q < nvalid) {
        cs = cs_arr[e + q];
        vv = val_arr[e + q];
    }
    f32x4 x[8];
    float vj[8];
    int sj[8];
#define SWEEP_FETCH(J)                                                                              \
    {                                                                                               \
        const int cj = swz_i<AND | ((J) << 5)>(cs);                                                 \
        vj[J] = swz_f<AND | ((J) << 5)>(vv);                                                        \
        sj[J] = (int)((unsigned)cj >> SWEEP_COL_BITS);                                              \
        const uint32_t boff = (uint32_t)(cj & ((1 << SWEEP_COL_BITS) - 1)) * (uint32_t)(D * 4) + (uint32_t)(sl * 16); \
        if (!PRED || (J) < nvalid) x[J] = *reinterpret_cast<const f32x4 *>(xb + boff);              \
    }
    SWEEP_FETCH(0) SWEEP_FETCH(1) SWEEP_FETCH(2) SWEEP_FETCH(3)
    SWEEP_FETCH(4) SWEEP_FETCH(5) SWEEP_FETCH(6) SWEEP_FETCH(7)
#undef SWEEP_FETCH
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (!PRED || j < nvalid) {
            // read-modify-write of the owner group's accumulator row (LDS operations of one wave
            // execute in order, nobody else touches this row).  ds_add_f32 would be one instruction
            // per float but measured ~200 clk per wave instruction on gfx950 -- 13x slower overall.
            f32x4 *p = reinterpret_cast<f32x4 *>(my + sj[j] * D);
            f32x4 t = *p;
            t[0] = fmaf(vj[j], x[j][0], t[0]);
            t[1] = fmaf(vj[j], x[j][1], t[1]);
            t[2] = fmaf(vj[j], x[j][2], t[2]);
            t[3] = fmaf(vj[j], x[j][3], t[3]);
            *p = t;
        }
    }
}

template <int D>
__global__ __launch_bounds__(1024) void spmm_sweep_kernel(SweepArgs a, SpmmArgs o) {
    extern __shared__ __attribute__((aligned(16))) float sweep_lds[];
    constexpr int LPG = D / 4;            // lanes per group: 8, 16, 32, 64
    constexpr int GPW = 64 / LPG;         // groups per wave
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / LPG, sl = lane % LPG;
    const int gl = wave * GPW + g;                                   // group inside the workgroup
    const int gid = blockIdx.x * (SWEEP_WPW * GPW) + gl;
    float *my = sweep_lds + (size_t)gl * (SWEEP_SPG * D) + sl * 4;  // this lane's 4 floats of slot 0
    {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < SWEEP_SPG; ++j) *reinterpret_cast<f32x4 *>(my + j * D) = z;
    }
    int e = 0, end = 0;
    if (gid < a.n_groups) {
        e = a.s_start[gid];
        end = e + a.s_len[gid];
    }
    const char *xb = reinterpret_cast<const char *>(o.X);
    while (true) {
        const int nvalid = end - e;                  // uniform inside a group
        if (!__any(nvalid > 0)) break;
        if (__all(nvalid >= 8)) {
            sweep_batch<D, false>(a, xb, my, e, 8, sl, a.cs, a.val);
        } else if (nvalid > 0) {
            sweep_batch<D, true>(a, xb, my, e, nvalid, sl, a.cs, a.val);
        }
        e += 8;
    }
    // every group only touched its own slots and LDS operations of a wave complete in order:
    // no barrier is needed before reading the accumulators back
    if (gid < a.n_groups) {
#pragma unroll 1
        for (int j = 0; j < SWEEP_SPG; ++j) {
            const int dst = a.g_dst[(size_t)gid * SWEEP_SPG + j];
            if (dst == SWEEP_UNUSED) continue;
            const f32x4 r = *reinterpret_cast<const f32x4 *>(my + j * D);
            float acc[4] = {r[0], r[1], r[2], r[3]};
            if (dst < 0) {
                vec_store<4>(o.partial + (size_t)(~dst) * D + sl * 4, acc);
                continue;
            }
            const size_t base = (size_t)dst * D + sl * 4;
            if (o.noise) {
                float n[4];
                vec_load<4>(n, o.noise + base);
                float ss = n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3];
                ss = group_sum<LPG>(ss);
                const float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = acc[i] + ((n[i] / nrm) * sign_f(acc[i])) * o.eps;
            }
            if (o.Y) vec_store<4>(o.Y + base, acc);
            if (o.acc_out) {
                float t[4];
                vec_load<4>(t, o.acc_in + base);
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] += acc[i];
                vec_store<4>(o.acc_out + base, t);
            }
        }
    }
}

// one wavefront per stream: kept entries packed to the front, s_len_out = kept count
__global__ __launch_bounds__(256) void sweep_compact_kernel(const int32_t *s_start, const int32_t *s_len,
                                                            int n_groups, const int32_t *cs, const float *val,
                                                            const int32_t *edge_map, const uint8_t *keep,
                                                            float scale, int32_t *cs_out, float *val_out,
                                                            int32_t *s_len_out) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + wave_in_block();
    if (gid >= n_groups) return;
    const int base = s_start[gid];
    const int len = s_len[gid];
    int kept = 0;
    for (int e0 = 0; e0 < len; e0 += 64) {
        const int e = e0 + lane;
        bool kp = false;
        int cc = 0;
        float vv = 0.f;
        if (e < len) {
            kp = keep[edge_map[base + e]] != 0;
            cc = cs[base + e];
            vv = val[base + e] * scale;
        }
        const unsigned long long m = __ballot(kp);
        if (kp) {
            const int pos = base + kept + __popcll(m & ((1ull << lane) - 1ull));
            cs_out[pos] = cc;
            val_out[pos] = vv;
        }
        kept += __popcll(m);
    }
    if (lane == 0) s_len_out[gid] = kept;
}

// ---- host launchers -----------------------------------------------------------------------
template <int D, bool BIG>
static int launch_spmm_big(const SpmmArgs &a, const sslrec_csr_t *A, hipStream_t st) {
    constexpr int GPW = 64 / (D / 4);
    const int waves = (a.n_waves + GPW - 1) / GPW;
    const int blocks = (waves + 3) / 4;
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

template <int D>
static int launch_sweep(const SweepArgs &a, const SpmmArgs &o, const sslrec_sweep_t *A, hipStream_t st) {
    constexpr int LPG = D / 4, GPW = 64 / LPG;
    const size_t lds_bytes = (size_t)SWEEP_WPW * GPW * SWEEP_SPG * D * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_sweep_kernel<D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (A->n_wg > 0) {
        hipLaunchKernelGGL((spmm_sweep_kernel<D>), dim3(A->n_wg), dim3(1024), lds_bytes, st, a, o);
        SSLREC_LAUNCH_CHECK();
    }
    if (A->n_long > 0) {
        hipLaunchKernelGGL((spmm_long_reduce_kernel<D>), dim3((A->n_long + 3) / 4), dim3(256), 0, st, o,
                           A->long_row, A->long_ptr, A->n_long);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sslrec_spmm_sweep_f32(const sslrec_sweep_t *A, const int32_t *cs_override,
                                     const float *val_override, const int32_t *s_len_override, const float *X,
                                     float *Y, const sslrec_epilogue_t *epi, float *partial_ws, void *stream) {
    if (!A || !X) return SSLREC_E_BADARG;
    if (!Y && !(epi && epi->acc_out)) return SSLREC_E_BADARG;
    if (A->n_slots > 0 && !partial_ws) return SSLREC_E_BADARG;
    if (epi && ((epi->acc_in == nullptr) != (epi->acc_out == nullptr))) return SSLREC_E_BADARG;
    if ((unsigned long long)A->n_cols * (unsigned long long)(A->d * 4) >= (1ull << 32)) return SSLREC_E_BADARG;
    if (A->n_cols >= (1 << SWEEP_COL_BITS)) return SSLREC_E_BADARG;
    SweepArgs a;
    a.s_start = A->s_start;
    a.s_len = s_len_override ? s_len_override : A->s_len;
    a.cs = cs_override ? cs_override : A->cs;
    a.val = val_override ? val_override : A->val;
    a.g_dst = A->g_dst;
    a.n_groups = A->n_groups;
    SpmmArgs o = {};
    o.X = X;
    o.Y = Y;
    o.partial = partial_ws;
    o.noise = epi ? epi->noise : nullptr;
    o.eps = epi ? epi->eps : 0.f;
    o.acc_in = epi ? epi->acc_in : nullptr;
    o.acc_out = epi ? epi->acc_out : nullptr;
    hipStream_t st = (hipStream_t)stream;
    switch (A->d) {
        case 32: return launch_sweep<32>(a, o, A, st);
        case 64: return launch_sweep<64>(a, o, A, st);
        case 128: return launch_sweep<128>(a, o, A, st);
        case 256: return launch_sweep<256>(a, o, A, st);
        default: return SSLREC_E_BADARG;
    }
}

extern "C" int sslrec_sweep_compact(const sslrec_sweep_t *A, const int32_t *edge_map, const uint8_t *keep,
                                    float scale, int32_t *cs_out, float *val_out, int32_t *s_len_out,
                                    void *stream) {
    if (!A || !edge_map || !keep || !cs_out || !val_out || !s_len_out) return SSLREC_E_BADARG;
    const int blocks = (A->n_groups + 3) / 4;
    if (blocks > 0) {
        hipLaunchKernelGGL(sweep_compact_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A->s_start,
                           A->s_len, A->n_groups, A->cs, A->val, edge_map, keep, scale, cs_out, val_out,
                           s_len_out);
        SSLREC_LAUNCH_CHECK();
    }
    return 0;
}
