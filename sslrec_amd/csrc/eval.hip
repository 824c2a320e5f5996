// All-rank evaluation and negative sampling on the device.
//
// 1. sslrec_eval_topk_f32 replaces, per evaluation batch, `full_predict` + `_mask_predict` + `t.topk`
//    (reference models/general_cf/lightgcn.py:58-66, models/base_model.py:35-36, trainer/metrics.py:99-103): the
//    reference ships a dense [B, I] int64 train mask from the host (750 MB per batch of 1024 at amazon-book size),
//    materializes the [B, I] score matrix, rewrites it three times and sorts it.  Here a wave keeps 32 users resident
//    as the B operand of exact-fp32 MFMA tiles (v_mfma_f32_32x32x2_f32, scores TRANSPOSED so that one lane sees 16
//    items of ONE user per tile, as in infonce.hip) and streams the item table once.  A lane meets its user's items in
//    ascending order, so the train mask is a MERGE against the user's sorted train row (tiles without a train item of
//    any of the wave's users skip it).  Scores that beat the user's threshold go to a per-user key buffer in LDS; a full
//    buffer is ranked by the wave, cut to the k best, and the threshold rises to the k-th (see eval_topk_kernel).  A
//    second small kernel merges the per-split lists: descending by score, ties by ascending item id.  Nothing of size
//    B x I ever exists.
// 2. sslrec_sample_negs replaces PairwiseTrnData.sample_negs (data_utils/datasets_general_cf.py:13-20: one
//    `np.random.randint(item_num)` per interaction, redrawn while the pair is a train interaction -- a Python loop
//    with dok lookups, 2.2 us per edge): one lane per interaction, Philox draws (philox.h), binary search in the
//    user's sorted train row.  Same distribution, different random stream.
#include "common.h"
#include "philox.h"
#include <algorithm>
#include <stdlib.h>

typedef float ev_f32x4 __attribute__((ext_vector_type(4)));
typedef float ev_f32x16 __attribute__((ext_vector_type(16)));

#ifndef EV_FEW_LANES
#define EV_FEW_LANES 6                  // at most this many lanes with a candidate: scalar handling (0 = always the wave-wide form)
#endif
#define EVAL_KMAX 64
#define EVAL_SLACK 2                    // a user's buffer is cut back when fewer slots than this are free

__device__ __forceinline__ int ev_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// k-major operand fragment of table rows: lane (r = lane & 31, h = lane >> 5) holds row `row`, elements [h*D/2, (h+1)*D/2)
template <int D>
__device__ __forceinline__ void ev_load_frag(float (&f)[D / 2], const float *__restrict__ base, int64_t row, int lane) {
    const ev_f32x4 *p = reinterpret_cast<const ev_f32x4 *>(base + (size_t)row * D + (lane >> 5) * (D / 2));
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
        const ev_f32x4 v = p[q];
        f[4 * q + 0] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3];
    }
}

// ---- "h3" score tiles (round 5; the arithmetic of infonce_x3.inc's h3 mode): both tables as TWO fp16 planes h0 = fp16(s x),
// h1 = fp16(s x - h0) with a per-table power-of-two scale s chosen from the table's largest magnitude on the device, a score tile =
// the three v_mfma_f32_32x32x16_f16 terms h0h0 + h0h1 + h1h0 (12 matrix instructions of 32 cycles at d = 64 where the exact-fp32 tile
// takes 32 of 64 cycles), scaled back by 1 / (s_u s_i).  22-bit operands: score errors ~2e-7 of the score's scale -- evaluation ranks
// scores, and the parity tests hold the ranks wherever two scores differ by more than 1e-5.  SSLREC_EVAL_PRECISION=fp32 keeps the
// exact-fp32 tiles.
typedef __bf16 ev_b8 __attribute__((ext_vector_type(8)));          // 16-byte container of 8 plane elements
typedef _Float16 ev_h8 __attribute__((ext_vector_type(8)));
typedef unsigned short ev_u16;

template <int D, bool H3> struct EvFrag;
template <int D> struct EvFrag<D, false> { float f[D / 2]; };
template <int D> struct EvFrag<D, true> { ev_b8 p[2][D / 16]; };

struct EvPlanes { const ev_u16 *u0, *u1, *i0, *i1; const float *inv_scale; };      // user planes in POSITION order, item planes by item id

template <int D, bool H3>
__device__ __forceinline__ void ev_load(EvFrag<D, H3> &f, const float *__restrict__ base, const ev_u16 *__restrict__ q0,
                                        const ev_u16 *__restrict__ q1, int64_t row, int lane) {
    if constexpr (H3) {
        const size_t at = (size_t)row * D + (lane >> 5) * (D / 2);
        const ev_b8 *a = reinterpret_cast<const ev_b8 *>(q0 + at), *b = reinterpret_cast<const ev_b8 *>(q1 + at);
#pragma unroll
        for (int q = 0; q < D / 16; ++q) { f.p[0][q] = a[q]; f.p[1][q] = b[q]; }
    } else {
        ev_load_frag<D>(f.f, base, row, lane);
    }
}

// s[item][user] of a 32 x 32 tile
template <int D, bool H3>
__device__ __forceinline__ ev_f32x16 ev_dot(const EvFrag<D, H3> &it, const EvFrag<D, H3> &us) {
    ev_f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
    if constexpr (H3) {
#define EV_TERM(I, J)                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < D / 16; ++q)                                                                         \
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ev_h8, it.p[I][q]), __builtin_bit_cast(ev_h8, us.p[J][q]), s, 0, 0, 0);
        EV_TERM(0, 1) EV_TERM(1, 0) EV_TERM(0, 0)
#undef EV_TERM
    } else {
#pragma unroll
        for (int kk = 0; kk < D / 2; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(it.f[kk], us.f[kk], s, 0, 0, 0);
    }
    return s;
}

// largest magnitude of (the gathered rows of) a table, as the bits of a non-negative float (they order like unsigned integers)
// (d = 32 / 64 / 128: a power of two -- a 64-bit division per element made the two preparation passes cost more than they saved on a
// batch of 1024 users)
__global__ __launch_bounds__(256) void ev_maxabs_kernel(const float *__restrict__ src, const int64_t *__restrict__ idx, int n, int d,
                                                        unsigned *__restrict__ out_bits) {
    float m = 0.f;
    const size_t total = (size_t)n * d;
    const int ld = 31 - __clz(d);
    for (size_t e4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; e4 < total; e4 += (size_t)gridDim.x * 1024) {
        const size_t r = e4 >> ld, c = e4 & (size_t)(d - 1);
        const ev_f32x4 v = *reinterpret_cast<const ev_f32x4 *>(src + (size_t)(idx ? idx[r] : (int64_t)r) * d + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));
}

// planes of (the gathered rows of) a table: scale = the power of two that brings the largest magnitude to [2^12, 2^13)
__global__ __launch_bounds__(256) void ev_planes_kernel(const float *__restrict__ src, const int64_t *__restrict__ idx, int n, int d,
                                                        const unsigned *__restrict__ max_bits, ev_u16 *__restrict__ p0, ev_u16 *__restrict__ p1,
                                                        float *__restrict__ scale_out) {
    const float m = __uint_as_float(max_bits[0]);
    const float scale = (m > 0.f && m < 3.0e38f) ? exp2f(floorf(log2f(8192.f / m))) : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = scale;
    const size_t total = (size_t)n * d;
    const int ld = 31 - __clz(d);
    typedef ev_u16 ev_u16x4 __attribute__((ext_vector_type(4)));
    for (size_t e4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; e4 < total; e4 += (size_t)gridDim.x * 1024) {
        const size_t r = e4 >> ld, c = e4 & (size_t)(d - 1);
        const ev_f32x4 v = *reinterpret_cast<const ev_f32x4 *>(src + (size_t)(idx ? idx[r] : (int64_t)r) * d + c);
        ev_u16x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = v[i] * scale;
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)(x - (float)h);
            hi[i] = __builtin_bit_cast(ev_u16, h);
            lo[i] = __builtin_bit_cast(ev_u16, l);
        }
        *reinterpret_cast<ev_u16x4 *>(p0 + e4) = hi;
        *reinterpret_cast<ev_u16x4 *>(p1 + e4) = lo;
    }
}

__global__ void ev_inv_scale_kernel(const float *scales, float *inv) { inv[0] = 1.f / (scales[0] * scales[1]); }

// is `item` one of the (sorted) train items of the user whose row is [lo, hi)?
__device__ __forceinline__ bool ev_seen(const int64_t *__restrict__ col, int64_t lo, int64_t hi, int64_t item) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t c = col[mid];
        if (c == item) return true;
        if (c < item) lo = mid + 1; else hi = mid;
    }
    return false;
}

// A candidate is one 64-bit key: the score's bits made order-preserving in the high word, ~item in the low word, so that
// "larger key" == "higher score, ties to the smaller item id" (the order of the reference's topk up to ties) and keys are
// unique.  Key 0 (a negative NaN) is "empty" and loses to everything.
__device__ __forceinline__ uint64_t ev_key(float v, int item) {
    uint32_t b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((uint64_t)b << 32) | (uint32_t)(~(uint32_t)item);
}
__device__ __forceinline__ float ev_key_val(uint64_t key) {
    uint32_t b = (uint32_t)(key >> 32);
    b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
    return __uint_as_float(b);
}
__device__ __forceinline__ int ev_key_item(uint64_t key) { return (int)(~(uint32_t)key); }

__device__ __forceinline__ void ev_wave_sync() {          // LDS traffic between the lanes of ONE wave
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The whole wave ranks the n (<= C) keys of one user's buffer (rank = number of larger keys; keys are unique) and keeps
// the k best, in rank order, at the front.  `dst` == nullptr: in place, *thr = the k-th best; otherwise the ranked keys go
// to dst[0..k) (global memory), missing ones as 0 -- and so are keys below `floor_key`, a lower bound of the user's k-th best overall
// that rose after they were taken (the merge then finds a few keys per split instead of k).
template <int C>
__device__ __forceinline__ void ev_rank_keep(uint64_t *kb, int *cnt, uint64_t *thr, int k, int lane, uint64_t *dst, uint64_t floor_key = 0ull,
                                             unsigned *pub_slot = nullptr, int pub_m = 0) {
    constexpr int PER = C / 64;
    const int n = min(__builtin_amdgcn_readfirstlane(*cnt), C);      // the counter may have run past a full buffer
    uint64_t mine[PER];
    int rank[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        mine[p] = (lane + 64 * p < n) ? kb[lane + 64 * p] : 0ull;
        rank[p] = 0;
    }
    // uniform addresses: LDS broadcasts, 8 in flight (slots at and beyond n hold stale keys: they are masked out).  (Passing
    // the keys round as scalars with v_readlane instead was measured slower.)
    for (int j0 = 0; j0 < n; j0 += 8) {
        uint64_t kj[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kj[i] = kb[(j0 + i) & (C - 1)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (j0 + i >= n) kj[i] = 0ull;
#pragma unroll
            for (int p = 0; p < PER; ++p) rank[p] += (kj[i] > mine[p]) ? 1 : 0;
        }
    }
    ev_wave_sync();
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        if (lane + 64 * p < n && rank[p] < k) {
            if (dst) dst[rank[p]] = mine[p] >= floor_key ? mine[p] : 0ull;
            else {
                kb[rank[p]] = mine[p];
                if (rank[p] == k - 1) *thr = mine[p];
                // (few splits per user: the split's m-th best so far is published -- m items of THIS split are at or above it; see `share_few`)
                if (pub_slot && rank[p] == pub_m - 1)
                    __hip_atomic_store(pub_slot, (unsigned)(mine[p] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (dst) {
        if (lane < k && lane >= n) dst[lane] = 0ull;
    } else if (lane == 0) {
        *cnt = min(n, k);
    }
    ev_wave_sync();
}

// One wave = 32 users (the MFMA B operand, resident) x the items of one split.  Per user: a buffer of C keys in LDS fed
// by both of the user's lanes (LDS atomic counter) with every score that beats the user's threshold; when a buffer is
// nearly full the wave ranks it, keeps the k best and raises the threshold to the k-th.  A threshold is stale between
// two cuts, which only admits some extra candidates on top of the k ln(I/k) an exact running list would see -- against
// that, a score costs two compares unless it is a candidate.  C = 64 for k <= 48: 16 KB of LDS per wave, two 4-wave
// blocks per CU, so one wave's candidate handling runs under the other's MFMA chain.
// part_key [n_users][n_split][k]
// MODE: how the item splits of a user share what they know -- 0 the running maximum of their own k-th bests (gthr), 1 also the k-th
// largest of their published best scores (n_split >= k, `share`), 2 also the j-th largest of their published m-th bests (n_split <= 4,
// `share_few`); a template parameter because the D = 64 kernels have no register to spare for code they do not run
// STAGE (round 6, opt-in: measured slower, see ev_stage_enabled; h3, d <= 64, C = 64): a workgroup of NW = 8 waves (256 users) stages every 32-item tile ONCE in LDS (LDS-DMA, two
// buffers, one barrier per tile) and all its waves read the operand fragments from there -- in the register form every wave fetches the
// tile for itself from the L2: 1,646 user groups x 23 MB of item planes = 38 GB through the L2 -> L1 path for all amazon-book users,
// which is what the launch waits for (EXPERIMENTS C.8b), and two register sets of 32 registers per lane.  One workgroup per CU (128 KiB
// of key buffers + 16 KiB of stage).  A wave without users still copies its slab and meets the barriers.
template <int D, int C, bool H3, int MODE, int NW = 4, bool STAGE = false>
__global__ __launch_bounds__(NW * 64, 2) void eval_topk_kernel(const float *__restrict__ UE, const int64_t *__restrict__ users, int n_users,
                                                        const float *__restrict__ IE, int n_items, EvPlanes pl,
                                                        const int64_t *__restrict__ trn_rowptr, const int64_t *__restrict__ trn_col,
                                                        int k, int n_ugroup, int items_per_split, int n_split, int cut_at,
                                                        uint64_t *__restrict__ part_key, unsigned long long *__restrict__ gthr,
                                                        unsigned *__restrict__ pub, int pub_m, int pub_j) {
    extern __shared__ uint64_t ev_lds[];                  // [4 waves][32 users][C] keys, [4][32] thresholds, [4][32] counts
    constexpr int HALF = D / 2;
    const int lane = threadIdx.x & 63, h = lane >> 5, ur = lane & 31, w = wave_in_block();
    const int ug = blockIdx.x % n_ugroup, split = blockIdx.x / n_ugroup;
    const int u0 = (ug * NW + w) * 32;
    const bool active = u0 < n_users;
    if (!STAGE && !active) return;
    uint64_t *keys = ev_lds + (size_t)w * 32 * C;
    uint64_t *thr_l = ev_lds + (size_t)NW * 32 * C + w * 32;
    int *cnt_l = reinterpret_cast<int *>(ev_lds + (size_t)NW * 32 * C + NW * 32) + w * 32;
    ev_b8 *stage = reinterpret_cast<ev_b8 *>(ev_lds + (size_t)NW * 32 * C + NW * 32 + NW * 16);      // [2][2 planes x D/16 slabs][64 lanes] (STAGE)
    if (lane < 32) { thr_l[lane] = 0ull; cnt_l[lane] = 0; }
    ev_wave_sync();
    const int upos = min(u0 + ur, n_users - 1);
    const int64_t uid = users ? users[upos] : (int64_t)upos;
    typedef EvFrag<D, H3> Frag;
    Frag e1;
    ev_load<D, H3>(e1, UE, pl.u0, pl.u1, H3 ? (int64_t)upos : uid, lane);
    float inv_scale = 1.f;
    if constexpr (H3) inv_scale = pl.inv_scale[0];
    const int64_t row_hi = trn_rowptr ? trn_rowptr[uid + 1] : 0;
    uint64_t thr_key = 0ull;                              // the user's k-th best key so far (0: fewer than k seen)
#ifdef EV_NO_CAND
    float thr_f = INFINITY;                               // experiment: no candidate ever (the MFMA + streaming floor)
#else
    float thr_f = -3.402823466e+38f;                      // its score: the cheap first test (masked scores are -inf)
#endif
    float run_best = -INFINITY;                           // the best score this lane has seen for its user (pub: the split's published top-1)
    uint32_t pub_last = 0u;
    const int j_begin = split * items_per_split;
    const int j_end = min(j_begin + items_per_split, n_items);
    // cursor into the user's sorted train row: n0 = its first item >= j_begin, n1 = the one after (one load of look-ahead,
    // so that stepping the cursor never waits for memory)
    int64_t cur = trn_rowptr ? trn_rowptr[uid] : 0;
    {
        int64_t lo = cur, hi = row_hi;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (trn_col[mid] < j_begin) lo = mid + 1; else hi = mid;
        }
        cur = lo;
    }
    int n0 = cur < row_hi ? (int)trn_col[cur] : 0x7fffffff;
    int n1 = cur + 1 < row_hi ? (int)trn_col[cur + 1] : 0x7fffffff;
    // (few splits: where a cut publishes the split's m-th best of user u0 + U; nullptr otherwise -- the many-splits form publishes from `share`)
    unsigned *const pub_base = pub ? pub + (size_t)u0 * n_split + split : nullptr;      // this split's slot of the wave's first user
#define PUB_SLOT(U) ((MODE == 2 && u0 + (U) < n_users) ? pub_base + (U) * n_split : (unsigned *)nullptr)
    // one tile of 32 items: `cur_frag` holds its rows, the rows of the next tile are fetched into `next_frag` meanwhile
    // (the loop below alternates two register sets, so nothing is copied)
    auto tile = [&](const Frag &cur_frag, Frag &next_frag, const int j0, const ev_b8 *st_cur = nullptr) {
        ev_f32x16 s;
        if constexpr (STAGE) {      // the tile's fragments come out of the workgroup's LDS stage: slab (plane k, q) at st_cur[(k * NI + q) * 64 + lane]
            constexpr int NI = D / 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#define EV_TERM_L(I, J)                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < NI; ++q)                                                                             \
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ev_h8, st_cur[((I) * NI + q) * 64 + lane]),              \
                                                   __builtin_bit_cast(ev_h8, e1.p[J][q]), s, 0, 0, 0);
            EV_TERM_L(0, 1) EV_TERM_L(1, 0) EV_TERM_L(0, 0)
#undef EV_TERM_L
        } else {
#ifndef EV_NO_LOAD      // (experiment: the loop without its item loads -- both register sets hold the split's first tile)
            if (j0 + 32 < j_end) ev_load<D, H3>(next_frag, IE, pl.i0, pl.i1, min(j0 + 32 + ur, n_items - 1), lane);
#endif
            s = ev_dot<D, H3>(cur_frag, e1);      // s[item][user]
        }
        if constexpr (H3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] *= inv_scale;
        }
        // (two accumulator chains over the halves of d instead of one chain of 32 dependent MFMAs: measured, no change -- 8.72 against
        // 8.80 ms for all users with the thresholds at +inf, EXPERIMENTS.md A.8: the chain is not what holds the matrix pipe at 33 %)
        if (__ballot(n0 < j0 + 32)) {                     // a train item of some user in this tile (rare): the lane that sees it masks it
            do {
                if (n0 < j0 + 32) {
                    const int o = n0 - j0;                // 0 .. 31; this lane holds the items with ((o >> 2) & 1) == h
                    const int rr = (((o >> 2) & 1) == h) ? (o & 3) + 4 * (o >> 3) : -1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (r == rr) ? -INFINITY : s[r];
                    ++cur;
                    n0 = n1;
                    n1 = cur + 1 < row_hi ? (int)trn_col[cur + 1] : 0x7fffffff;
                }
            } while (__ballot(n0 < j0 + 32));
        }
        if (j0 + 32 > j_end) {                            // the ragged last tile
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + ev_crow(r, h) >= j_end) s[r] = -INFINITY;
        }
        float best = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) best = fmaxf(best, s[r]);
        run_best = fmaxf(run_best, best);
        uint64_t cand_lanes = __ballot(best >= thr_f);
        if (!cand_lanes) return;
        // candidates: first the cheap test on every score, then the exact one (score, then item id)
        unsigned maybe = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) maybe |= (s[r] >= thr_f) ? 1u << r : 0u;
#if EV_FEW_LANES
        // FEW lanes with a candidate -- the usual tile once the thresholds have risen: one or two scores out of 1024 -- are handled one
        // after the other with SCALAR control: the lane's bit mask, its threshold and each candidate score come out with v_readlane
        // (the score through a uniform register index), the key is built and tested in scalar registers and one lane stores it; no
        // LDS atomic, no exchange between the user's two lanes.  The wave-wide form below costs the same ~300 instructions (16
        // wave-uniform tests, twice, plus two LDS round trips) whether the tile holds one candidate or a thousand: measured on the loop
        // without its item loads, 5.5 of 12.4 ms (EXPERIMENTS.md A.8).  It stays for the first tiles of a split, where most scores pass.
        if (__popcll(cand_lanes) <= EV_FEW_LANES) {
            uint64_t cut_users = 0ull;
            bool cut_any = false;
            do {
                const int L = __ffsll((unsigned long long)cand_lanes) - 1;
                cand_lanes &= cand_lanes - 1;
                unsigned m = (unsigned)__builtin_amdgcn_readlane((int)maybe, L);
                uint64_t tk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(thr_key >> 32), L) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)thr_key, L);
                const int u = L & 31, hh = L >> 5;
                int c = cnt_l[u];
                while (m) {
                    const int r = __ffs(m) - 1;
                    m &= m - 1;
                    const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s[r]), L));
                    if (!(v > -INFINITY)) continue;                     // masked (train item, ragged tile)
                    const uint64_t key = ev_key(v, j0 + (r & 3) + 8 * (r >> 2) + 4 * hh);
                    if (!(key > tk)) continue;
                    if (c >= C) {                                       // the buffer is full: cut it now, test against the new threshold
                        if (lane == 0) cnt_l[u] = c;
                        ev_wave_sync();
                        ev_rank_keep<C>(keys + u * C, cnt_l + u, thr_l + u, k, lane, nullptr, 0ull, PUB_SLOT(u), pub_m);
                        c = cnt_l[u];
                        tk = thr_l[u];
                        cut_any = true;
                        if (!(key > tk)) continue;
                    }
                    if (lane == 0) keys[u * C + c] = key;
                    ++c;
                }
                if (lane == 0) cnt_l[u] = c;
                if (c > cut_at) cut_users |= 1ull << u;
            } while (cand_lanes);
            if (!cut_users && !cut_any) return;
            ev_wave_sync();
            while (cut_users) {
                const int u = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)cut_users) - 1);
                cut_users &= cut_users - 1;
                ev_rank_keep<C>(keys + u * C, cnt_l + u, thr_l + u, k, lane, nullptr, 0ull, PUB_SLOT(u), pub_m);
            }
            uint64_t nk = thr_l[ur];
            if (gthr && u0 + ur < n_users) {                            // (shared thresholds: see the wave-wide form below)
                const unsigned long long seen = h == 0 ? atomicMax(gthr + u0 + ur, (unsigned long long)nk) : 0ull;
                const uint64_t other = max((uint64_t)seen, (uint64_t)__shfl_xor(seen, 32, 64));
                if (other > nk) nk = other;
            }
            if (nk != thr_key) { thr_key = nk; thr_f = ev_key_val(thr_key); }
            return;
        }
#endif
        const float thr_v = thr_key ? ev_key_val(thr_key) : -INFINITY;
        const int thr_item = ev_key_item(thr_key);
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (__ballot(maybe & (1u << r))) {
                // (unsigned: the published-maxima bound carries item id 0xffffffff, "every item of this score passes")
                const bool pass = s[r] > thr_v || (s[r] == thr_v && (unsigned)(j0 + ev_crow(r, h)) < (unsigned)thr_item);
                hits |= ((maybe & (1u << r)) && pass && s[r] > -INFINITY) ? 1u << r : 0u;
            }
        }
        // One LDS atomic per lane and round.  Keys that do not fit the user's buffer stay in `hits` and are offered again
        // after the buffer has been cut back to its k best (only the first tiles of a split, or scores arriving in
        // ascending order, overflow: a buffer is cut as soon as it holds more than `cut_at` keys).
        while (__ballot(hits != 0)) {
            int filled = 0;
            if (hits) {
                const int want = __popc(hits);
                const int base = atomicAdd(&cnt_l[ur], want);
                int at = base;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (__ballot(hits & (1u << r))) {
                        if (hits & (1u << r)) {
                            if (at < C) { keys[ur * C + at] = ev_key(s[r], j0 + ev_crow(r, h)); hits &= ~(1u << r); }
                            ++at;
                        }
                    }
                }
                filled = base + want;
            }
            filled = max(filled, __shfl_xor(filled, 32, 64));                   // the user's other lane
            uint64_t need = __ballot(filled > cut_at) & 0xffffffffull;
            if (!need) break;                                                   // nothing overflowed either
            ev_wave_sync();
            while (need) {
                const int u = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)need) - 1);
                need &= need - 1;
                ev_rank_keep<C>(keys + u * C, cnt_l + u, thr_l + u, k, lane, nullptr, 0ull, PUB_SLOT(u), pub_m);
            }
            uint64_t nk = thr_l[ur];
            if (gthr && u0 + ur < n_users) {
                // SHARED thresholds (item splits of the same users run on different CUs): a key below the k-th best of ANY split cannot
                // be among the user's k best overall, so the splits raise one running maximum per user in global memory (a device-scope
                // atomic max per cut) and adopt it -- a split no longer pays the k ln(n/k) start-up candidates of its own items once
                // another split has found k good ones.  Exact: the merge still sees every key that could make the final list.
                const unsigned long long seen = h == 0 ? atomicMax(gthr + u0 + ur, (unsigned long long)nk) : 0ull;
                const uint64_t other = max((uint64_t)seen, (uint64_t)__shfl_xor(seen, 32, 64));
                if (other > nk) nk = other;
            }
            if (nk != thr_key) {                                                // the threshold rose: drop what no longer beats it
                thr_key = nk;
                thr_f = ev_key_val(thr_key);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((hits & (1u << r)) && !(ev_key(s[r], j0 + ev_crow(r, h)) > thr_key)) hits &= ~(1u << r);
            }
        }
    };
    Frag fa, fb;
    if (!STAGE && j_begin < j_end) ev_load<D, H3>(fa, IE, pl.i0, pl.i1, min(j_begin + ur, n_items - 1), lane);
#ifdef EV_NO_LOAD
    ev_load<D, H3>(fb, IE, pl.i0, pl.i1, min(j_begin + ur, n_items - 1), lane);
#endif
    auto adopt = [&]() {                                   // pick up what the other splits have found (every 8 tiles)
        if (!gthr) return;
        const uint64_t g = __hip_atomic_load(gthr + min(u0 + ur, n_users - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g > thr_key) { thr_key = g; thr_f = ev_key_val(g); }
    };
    // MANY splits per user (a batch of 1024 users: 64 splits of 45 tiles): a split's own k-th best rises slowly -- it has to find k
    // good items among ITS 1,431 -- and the shared maximum of those (gthr) is no better than the best of them, so every split pays
    // k (1 + ln(items_per_split / k)) candidates and a cut per (C - k) of them: 64 x 183 candidates per user where one pass over all
    // items would see 350.  With n_split >= k there is a much better bound for free: every split publishes the best score it has
    // seen (pub, the order-preserving bits of the score cut to 18), and the k-th largest of the splits' maxima has k distinct items
    // at or above it -- a lower bound of the user's k-th best overall that is close to it already after the first tile, because the k
    // best of 64 x 32 items mostly sit in different splits.  The wave finds it per user with a radix search over the 64 lanes' values
    // (18 ballots).  Exact: a bound below the k-th best never rejects one of the k best; what it admits the merge sorts out.
    auto share = [&]() {
        const float b = fmaxf(run_best, __shfl_xor(run_best, 32, 64));
        const uint32_t ob = (uint32_t)(ev_key(b, 0) >> 32) & 0xFFFFC000u;
        // (a split that has seen no unmasked item publishes nothing: the cut bits of -inf would read back as a NaN)
        if (h == 0 && u0 + ur < n_users && b > -INFINITY && ob > pub_last)
            __hip_atomic_store(pub_base + ur * n_split, ob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b > -INFINITY) pub_last = ob;
        // the bound of user u0 + du is worked out by ONE of the user's splits (du = split, split + n_split, ...: a wave has 32 users) and
        // handed to the others through the shared threshold -- 2048 published values per wave and refresh would cost more than the cuts saved
        const int nu = min(32, n_users - u0);
        for (int du = split; du < nu; du += n_split) {
            const uint32_t v = lane < n_split
                                   ? __hip_atomic_load(pub + (size_t)(u0 + du) * n_split + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            uint32_t res = 0u;
            for (int bit = 31; bit >= 14; --bit) {
                const uint32_t cand = res | (1u << bit);
                if (__popcll(__ballot(v >= cand)) >= k) res = cand;
            }
            if (res && lane == 0) atomicMax(gthr + u0 + du, (unsigned long long)res << 32);
        }
        const uint64_t g = __hip_atomic_load(gthr + min(u0 + ur, n_users - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g > thr_key) { thr_key = g; thr_f = ev_key_val(g); }
    };
    // FEW splits per user (all 52,643 users: 3): a split's own k-th best among ITS third of the items sits near the user's 3k-th best
    // overall, so every split admits three times the candidates one list over all items would.  Each cut publishes the split's m-th
    // best, m = ceil(k / n_split) (ev_rank_keep); the j-th largest of the published values, j = ceil(k / m), has j x m >= k distinct
    // items at or above it: a bound near the k-th best of everything the splits have seen TOGETHER.  Every lane reads its user's
    // n_split <= 4 values at the refresh it does anyway (the loads travel together with the shared threshold's).
    auto share_few = [&]() {
        const size_t gi = min(u0 + ur, n_users - 1);
        const uint64_t g = __hip_atomic_load(gthr + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t v[4];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp)
            v[sp] = sp < n_split ? __hip_atomic_load(pub + gi * n_split + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        uint32_t sel = 0u;                                 // the pub_j-th largest (0 while fewer than pub_j splits have published)
#pragma unroll
        for (int a2 = 0; a2 < 4; ++a2) {
            int larger = 0;
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) larger += (v[b2] > v[a2] || (v[b2] == v[a2] && b2 < a2)) ? 1 : 0;
            if (larger == pub_j - 1) sel = v[a2];
        }
        uint64_t best = g;
        const uint64_t t1 = (uint64_t)sel << 32;
        if (t1 > best) {
            if (h == 0 && u0 + ur < n_users) atomicMax(gthr + gi, (unsigned long long)t1);
            best = t1;
        }
        if (best > thr_key) { thr_key = best; thr_f = ev_key_val(best); }
    };
    // tiles done when the thresholds are refreshed: every 8 without the published maxima; with them after every tile up to 4, every second up
    // to 16 (the published maxima of the other splits arrive a refresh late, the bound another refresh later)
    auto share_at = [](int t) { return t >= 1 && (t <= 4 || (t <= 16 && (t & 1) == 0) || (t & 7) == 0); };
    if constexpr (STAGE) {
        static_assert(!STAGE || (H3 && D <= 64 && MODE != 1), "the staged form: fp16 planes, d <= 64, few splits");
        constexpr int NI = D / 16, SLABS = 2 * NI;
        // wave w copies slab w of the tile (plane k = w / NI, 16-byte piece q = w % NI of every lane's half row) global -> LDS, no registers
        auto fetch = [&](int j0, int buf) {
            if (w < SLABS) {
                const int kq = w, kk = kq / NI, q = kq % NI;
                const int row = min(j0 + ur, n_items - 1);
                const ev_u16 *src = (kk == 0 ? pl.i0 : pl.i1) + (size_t)row * D + h * (D / 2) + 8 * q;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(stage + ((size_t)buf * SLABS + kq) * 64), 16, 0, 0);
            }
        };
        const int nt = (j_end - j_begin + 31) / 32;
        if (nt > 0) fetch(j_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int j0 = j_begin + 32 * t, buf = t & 1;
            if (t + 1 < nt) fetch(j0 + 32, buf ^ 1);         // lands while this tile is scored; nobody reads that buffer now
            if (active) {
                if ((t & 7) == 0) { if constexpr (MODE == 2) share_few(); else adopt(); }
                tile(fa, fa, j0, stage + (size_t)buf * SLABS * 64);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my piece of the next tile has landed; the barrier covers the other waves'
            __syncthreads();
        }
        if (!active) return;
    } else if constexpr (D <= 64) {
        for (int j0 = j_begin; j0 < j_end; j0 += 64) {
            const int t = (j0 - j_begin) >> 5;
            if constexpr (MODE == 1) { if (share_at(t)) share(); }
            else if ((t & 7) == 0) { if constexpr (MODE == 2) share_few(); else adopt(); }
            tile(fa, fb, j0);
            if constexpr (MODE == 1) { if (share_at(t + 1) && t + 1 <= 3) share(); }
            if (j0 + 32 < j_end) tile(fb, fa, j0 + 32);
        }
    } else {                                              // two register sets + two copies of the tile code cost the second wave per SIMD
        for (int j0 = j_begin; j0 < j_end; j0 += 32) {
            const int t = (j0 - j_begin) >> 5;
            if ((t & 7) == 0) { if constexpr (MODE == 2) share_few(); else adopt(); }      // (no published MAXIMA at d = 128: the refresh costs this kernel registers it does not have)
            tile(fa, fb, j0);
            fa = fb;
        }
    }
    adopt();                                              // (the last word of the other splits: the floor of what this one hands to the merge)
    ev_wave_sync();
    for (int u = 0; u < 32; ++u) {
        if (u0 + u >= n_users) break;
        const uint64_t fl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(thr_key >> 32), u) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)thr_key, u);
        ev_rank_keep<C>(keys + u * C, cnt_l + u, thr_l + u, k, lane, part_key + ((size_t)(u0 + u) * n_split + split) * k, fl);
    }
}

// one wave per user: the k best of its n_cand = n_split * k candidate keys, in order; empty keys come out as item -1
template <int PER>                                      // candidates per lane, at most (n_cand <= 64 * PER)
__global__ __launch_bounds__(256) void eval_topk_merge_kernel(const uint64_t *__restrict__ part_key, int n_users, int n_cand, int k,
                                                              int64_t *__restrict__ out_idx, float *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + wave_in_block();
    if (u >= n_users) return;
    uint64_t v[PER];
    const int per = (n_cand + 63) / 64;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = i * 64 + lane;
        v[i] = (i < per && c < n_cand) ? part_key[(size_t)u * n_cand + c] : 0ull;
    }
    if constexpr (PER >= 8) {
        // Many splits hand over mostly EMPTY slots (a split keeps only what beats the shared bound of the user's k-th best): the filled ones
        // are packed into LDS and every key counts the larger ones -- its rank is its place in the list.  (1024 users x 64 splits x k = 40:
        // 117 us for the k rounds of "largest of 2560" below.)
        constexpr int CAP = 512;
        __shared__ uint64_t cbuf[4][CAP];
        uint64_t *cb = cbuf[wave_in_block()];
        int n = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (i < per) {
                const bool nz = v[i] != 0ull;
                const uint64_t m = __ballot(nz);
                const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                if (nz && pos < CAP) cb[pos] = v[i];
                n += __popcll(m);
            }
        }
        if (n <= CAP) {
            ev_wave_sync();
            uint64_t mine[CAP / 64];
            int rank[CAP / 64];
#pragma unroll
            for (int q = 0; q < CAP / 64; ++q) {
                mine[q] = (lane + 64 * q < n) ? cb[lane + 64 * q] : 0ull;
                rank[q] = 0;
            }
            for (int j0 = 0; j0 < n; j0 += 8) {
                uint64_t kj[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) kj[i] = cb[(j0 + i) & (CAP - 1)];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (j0 + i >= n) kj[i] = 0ull;
#pragma unroll
                    for (int q = 0; q < CAP / 64; ++q)
                        if (64 * q < n) rank[q] += (kj[i] > mine[q]) ? 1 : 0;      // (uniform: a few dozen keys need one slot per lane)
                }
            }
#pragma unroll
            for (int q = 0; q < CAP / 64; ++q) {
                if (lane + 64 * q < n && rank[q] < k) {
                    out_idx[(size_t)u * k + rank[q]] = (int64_t)ev_key_item(mine[q]);
                    if (out_val) out_val[(size_t)u * k + rank[q]] = ev_key_val(mine[q]);
                }
            }
            if (lane >= n && lane < k) {                  // fewer than k candidates in all: the tail is "no item"
                out_idx[(size_t)u * k + lane] = -1;
                if (out_val) out_val[(size_t)u * k + lane] = -INFINITY;
            }
            return;
        }
    }
    for (int t = 0; t < k; ++t) {
        uint64_t bv = 0ull;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (i < per && v[i] > bv) bv = v[i];
        uint64_t wv = bv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t ov = __shfl_xor((unsigned long long)wv, o, 64);
            if (ov > wv) wv = ov;
        }
        if (lane == 0) {
            out_idx[(size_t)u * k + t] = wv ? (int64_t)ev_key_item(wv) : -1;
            if (out_val) out_val[(size_t)u * k + t] = wv ? ev_key_val(wv) : -INFINITY;
        }
        if (wv && bv == wv) {                            // keys are unique: exactly one lane owns the winner
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (v[i] == wv) v[i] = 0ull;
        }
    }
}

// Buffer size per user and item splits, from measurements on the amazon-book-shaped data (tools/eval_sweep.py, d = 64):
// C = 64 keys (two blocks per CU) beats C = 128 (one) for all users, 12.9 against 17.6 ms at k = 40; cutting a buffer only
// when it is full beats cutting early (cut at 44 / 56 / 62 keys: 14.2 / 12.8 / 12.7 ms); an item split that restarts its
// thresholds costs about a quarter of a full pass per block (round 3: 1 / 2 / 3 / 5 / 8 splits for all users: 12.7 / 14.0 /
// 14.0 / 16.2 / 16.9 ms; 1024 users: 16 / 32 / 48 splits 1.42 / 1.04 / 0.97 ms) -- since round 4 the splits of a user group
// share their thresholds (gthr), which is what makes a few splits pay for many users too (below).
// Round 5: with n_split >= k (a batch of <= 1024 users) the splits also publish their best scores and the k-th largest of those bounds the
// user's k-th best (`share` in the kernel): 1024 users, k = 40: 0.78 -> 0.36 ms; more splits for larger batches do not pay (profiles/r05/eval_split_sweep.jsonl).
static int ev_cap(int k) { return k <= 48 ? 64 : 128; }
static int ev_choose_split(int n_users, int n_items, int k) {
    const int n_ugroup = (n_users + 127) / 128;
    const int s_max = 4096 / k < 64 ? 4096 / k : 64;          // the merge kernel takes n_split * k <= 4096 candidates (64 per lane)
    const int tiles = (n_items + 31) / 32;
    if (const char *e = getenv("SSLREC_EVAL_SPLIT")) {        // experiments (tools/eval_variants.py)
        const int v = atoi(e);
        if (v >= 1 && v <= s_max && v <= tiles) return v;
    }
    int s = (512 + n_ugroup - 1) / n_ugroup;                  // two blocks per CU
    // Many users (412 groups for amazon-book's 52,643): one block per group leaves the chip's 512 block slots 80 % full with blocks of
    // equal length, so the launch lasts as long as a CU with two.  With the thresholds shared between the splits (gthr) a few splits no
    // longer cost what they save: at least two full rounds of blocks (splits 1 / 2 / 3 / 4 / 5 / 6 / 8 / 10 for all amazon-book users:
    // 11.95 / 11.78 / 10.72 / 11.43 / 11.66 / 11.05 / 11.20 / 11.59 ms, profiles/r04/eval_sweep.jsonl).
    if (n_ugroup >= 256) s = (1024 + n_ugroup - 1) / n_ugroup;
    if (s > s_max) s = s_max;
    if (s > tiles / 16) s = tiles / 16 > 1 ? tiles / 16 : 1;  // no split shorter than 16 tiles
    return s;
}

// the staged form (eval_topk_kernel<..., 8, true>): workgroups of 256 users, ONE per CU -- the number of item splits that wastes the
// least of the last round of 256 workgroup slots, among 3 .. 8 (blocks of a launch have about the same length)
// OPT-IN (SSLREC_EVAL_STAGE=1): measured NEGATIVE in round 6 (profiles/r06/eval_stage_ab.json, same box, alternating processes): all
// 52,643 users, k = 40, d = 64: 9.37-9.42 ms staged against 7.80-7.89 ms for the register form (d = 32: 8.6 against 5.5; 16,384 users:
// 4.4 against 3.2) at 3 .. 8 item splits.  The tile's fragments are shared, but eight waves now meet at a barrier after every tile while
// their candidate handling is data dependent, and the 128 KiB of key buffers leave room for ONE workgroup per CU.
static bool ev_stage_enabled() {
    static const bool on = [] { const char *e = getenv("SSLREC_EVAL_STAGE"); return e && e[0] == '1'; }();
    return on;
}
static bool ev_stage_applies(int n_users, int k) { return ev_stage_enabled() && n_users >= 2048 && ev_cap(k) == 64; }
static int ev_choose_split_staged(int n_users, int n_items, int k) {
    const int n_ug = (n_users + 255) / 256;
    const int tiles = (n_items + 31) / 32;
    if (const char *e = getenv("SSLREC_EVAL_SPLIT")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4 && v <= tiles) return v;      // (the few-splits forms take at most 4 published values; more: gthr only)
        if (v >= 1 && v <= 64 && v <= tiles && v * k <= 4096) return v;
    }
    int best = 1;
    double best_eff = 0.0;
    for (int sp = 1; sp <= 8; ++sp) {
        if (sp > 1 && tiles / sp < 16) break;
        const int blocks = n_ug * sp;
        if (sp < 3 && blocks < 512 && n_ug * 8 >= 512) continue;      // at least two rounds of work per CU when the users allow it
        const double eff = (double)blocks / (double)(((blocks + 255) / 256) * 256);
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    return best;
}

static size_t ev_lists_bytes(int n_users, int n_items, int k) {
    size_t ns = ev_choose_split(n_users, n_items, k);
    if (ev_stage_applies(n_users, k)) ns = std::max<size_t>(ns, (size_t)ev_choose_split_staged(n_users, n_items, k));      // (sized for either form)
    // candidate lists + shared thresholds + the splits' published maxima (4 bytes per user and split, used when n_split >= k)
    return (size_t)n_users * ns * k * 8 + (size_t)n_users * 8 + (((size_t)n_users * ns * 4 + 7) & ~(size_t)7);
}

extern "C" size_t sslrec_eval_topk_ws_bytes(int32_t n_users, int32_t n_items, int32_t k) {
    if (n_users <= 0 || n_items <= 0 || k <= 0 || k > EVAL_KMAX) return 0;
    // + 64 bytes of scales + the fp16 planes of both tables (two planes of 2-byte elements; sized for d = 128, the widest kernel)
    return ev_lists_bytes(n_users, n_items, k) + 64 + ((size_t)n_users + (size_t)n_items) * 128 * 4;
}

static bool share_few_enabled() { const char *e = getenv("SSLREC_EVAL_SHARE_FEW"); return e && e[0] == '1'; }

extern "C" int sslrec_eval_topk_f32(const float *UE, const int64_t *users, int32_t n_users, const float *IE, int32_t n_items,
                                    int32_t d, const int64_t *trn_rowptr, const int64_t *trn_col, int32_t k, void *ws,
                                    int64_t *out_idx, float *out_val, void *stream) {
    if (!UE || !IE || n_users <= 0 || n_items <= 0 || k <= 0 || k > EVAL_KMAX || !ws || !out_idx || (d != 32 && d != 64 && d != 128) ||
        ((trn_rowptr == nullptr) != (trn_col == nullptr)) || ((uintptr_t)ws & 7))
        return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    // h3 pays a pass over both tables for its planes: from 2048 users on it wins (all 52,643 amazon-book users 7.9 against 9.7 ms, a
    // batch of 1024 users 0.93 against 0.77 ms with the first version of the preparation: profiles/r05/eval_h3.json); SSLREC_EVAL_PRECISION
    // = h3 / fp32 forces either
    static const int forced = [] { const char *e = getenv("SSLREC_EVAL_PRECISION"); return !e ? 0 : (e[0] == 'f' ? 2 : 1); }();
    const bool h3 = forced == 1 || (forced == 0 && n_users >= 2048);
    const bool staged = h3 && d <= 64 && ev_stage_applies(n_users, k);      // round 6: the item tile staged once per 8-wave workgroup in LDS
    const int n_ugroup = staged ? (n_users + 255) / 256 : (n_users + 127) / 128;
    int n_split = staged ? ev_choose_split_staged(n_users, n_items, k) : ev_choose_split(n_users, n_items, k);
    const int cap0 = ev_cap(k);
    const int cut_at = cap0 - EVAL_SLACK;                    // cut a buffer back to its k best when it is (nearly) full
    const int items_per_split = ((n_items + n_split - 1) / n_split + 31) / 32 * 32;
    uint64_t *part_key = (uint64_t *)ws;
    unsigned long long *gthr = nullptr;                     // one running threshold per user, shared by its item splits
    unsigned *pub = nullptr;                                // the splits' published maxima (n_split >= k: see `share` in the kernel)
    int pub_m = 0, pub_j = 0;                               // ... or their m-th bests (few splits: `share_few`)
    if (n_split > 1) {
        gthr = (unsigned long long *)(part_key + (size_t)n_users * n_split * k);
        static const bool no_top1 = [] { const char *e = getenv("SSLREC_EVAL_SHARE_TOP1"); return e && e[0] == '0'; }();      // A/B measurements
        if (n_split >= k && n_split <= 64 && d <= 64 && !no_top1 && !staged) pub = (unsigned *)(gthr + n_users);
        // few splits: the m-th best of every split, m = ceil(k / n_split).  OPT-IN (SSLREC_EVAL_SHARE_FEW=1, read per call): measured with
        // all 52,643 amazon-book users (3 splits, k = 40) 8.16-8.17 ms with it against 7.82-7.88 without, 32,768 users 5.42-5.45 against
        // 5.27-5.30, 16,384 users 2.99 against 3.28-3.29 (profiles/r05/eval_all_users.jsonl) -- at three or four splits the candidates
        // are not what the launch waits for (the item rows of the next tile are), so a bound twice as tight buys nothing there
        else if (n_split < k && n_split <= 4 && share_few_enabled()) {
            pub = (unsigned *)(gthr + n_users);
            pub_m = (k + n_split - 1) / n_split;
            pub_j = (k + pub_m - 1) / pub_m;
        }
        hipError_t e = hipMemsetAsync(gthr, 0, (size_t)n_users * 8 + (pub ? (size_t)n_users * n_split * 4 : 0), st);
        if (e != hipSuccess) return (int)e;
    }
    const int cap = ev_cap(k);
    const size_t lds = (size_t)4 * 32 * cap * 8 + 4 * 32 * 8 + 4 * 32 * 4;
    int ev_dev = 0;
    if (hipGetDevice(&ev_dev) != hipSuccess || ev_dev < 0 || ev_dev >= 64) return SSLREC_E_BADARG;
    EvPlanes pl = {};
    if (h3) {      // scales from the tables' largest magnitudes, then the planes: five small launches (the tables are read twice)
        char *base = (char *)ws + ev_lists_bytes(n_users, n_items, k);
        unsigned *mx = (unsigned *)base;                       // [0] users, [1] items
        float *scales = (float *)(base + 8), *inv = (float *)(base + 16);
        ev_u16 *up0 = (ev_u16 *)(base + 64), *up1 = up0 + (size_t)n_users * d;
        ev_u16 *ip0 = up1 + (size_t)n_users * d, *ip1 = ip0 + (size_t)n_items * d;
        hipError_t e = hipMemsetAsync(mx, 0, 8, st);
        if (e != hipSuccess) return (int)e;
        auto grid = [](size_t n) { const size_t b = (n / 4 + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); };      // (4 elements per thread)
        hipLaunchKernelGGL(ev_maxabs_kernel, dim3(grid((size_t)n_users * d)), dim3(256), 0, st, UE, users, n_users, d, mx);
        hipLaunchKernelGGL(ev_maxabs_kernel, dim3(grid((size_t)n_items * d)), dim3(256), 0, st, IE, (const int64_t *)nullptr, n_items, d, mx + 1);
        hipLaunchKernelGGL(ev_planes_kernel, dim3(grid((size_t)n_users * d)), dim3(256), 0, st, UE, users, n_users, d, mx, up0, up1, scales);
        hipLaunchKernelGGL(ev_planes_kernel, dim3(grid((size_t)n_items * d)), dim3(256), 0, st, IE, (const int64_t *)nullptr, n_items, d, mx + 1, ip0, ip1,
                           scales + 1);
        hipLaunchKernelGGL(ev_inv_scale_kernel, dim3(1), dim3(1), 0, st, scales, inv);
        SSLREC_LAUNCH_CHECK();
        pl = EvPlanes{up0, up1, ip0, ip1, inv};
    }
#define EV_GO3(DD, CC, HH, MM)                                                                                                \
    {                                                                                                                     \
        static bool attr_set[64] = {};      /* per instantiation and device; the call is slow on the host */               \
        if (!attr_set[ev_dev]) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void *)eval_topk_kernel<DD, CC, HH, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)((size_t)4 * 32 * (CC) * 8 + 4 * 32 * 8 + 4 * 32 * 4));                                   \
            if (e != hipSuccess) return (int)e;                                                                           \
            attr_set[ev_dev] = true;                                                                                      \
        }                                                                                                                 \
        hipLaunchKernelGGL((eval_topk_kernel<DD, CC, HH, MM>), dim3(n_ugroup * n_split), dim3(256), lds, st, UE, users, n_users, IE, n_items, pl, \
                           trn_rowptr, trn_col, k, n_ugroup, items_per_split, n_split, cut_at, part_key, gthr, pub, pub_m, pub_j); \
    }
#define EV_GO2(DD, CC, HH)                                                                                                    \
    {      /* (a launch in every case: the published-maxima form exists for d <= 64 only and `pub` is never set for it at d = 128) */ \
        if (pub && pub_m > 0) EV_GO3(DD, CC, HH, 2)                                                                             \
        else if (pub && DD <= 64) { if constexpr (DD <= 64) EV_GO3(DD, CC, HH, 1) }                                              \
        else EV_GO3(DD, CC, HH, 0)                                                                                               \
    }
#define EV_GO_STAGED(DD, MM)                                                                                                   \
    {                                                                                                                     \
        constexpr size_t lds_s = (size_t)8 * 32 * 64 * 8 + 8 * 32 * 8 + 8 * 32 * 4 + (size_t)2 * 2 * ((DD) / 16) * 1024;   \
        static bool attr_set[64] = {};                                                                                    \
        if (!attr_set[ev_dev]) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void *)eval_topk_kernel<DD, 64, true, MM, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s); \
            if (e != hipSuccess) return (int)e;                                                                           \
            attr_set[ev_dev] = true;                                                                                      \
        }                                                                                                                 \
        hipLaunchKernelGGL((eval_topk_kernel<DD, 64, true, MM, 8, true>), dim3(n_ugroup * n_split), dim3(512), lds_s, st, UE, users, n_users, IE, n_items, pl, \
                           trn_rowptr, trn_col, k, n_ugroup, items_per_split, n_split, cut_at, part_key, gthr, pub, pub_m, pub_j); \
    }
#define EV_GO(DD, CC) { if (h3) EV_GO2(DD, CC, true) else EV_GO2(DD, CC, false) }
    if (staged) {
        if (pub && pub_m > 0) { if (d == 32) EV_GO_STAGED(32, 2) else EV_GO_STAGED(64, 2) }
        else { if (d == 32) EV_GO_STAGED(32, 0) else EV_GO_STAGED(64, 0) }
    } else
    if (cap == 64) { if (d == 32) EV_GO(32, 64) else if (d == 64) EV_GO(64, 64) else EV_GO(128, 64) }
    else { if (d == 32) EV_GO(32, 128) else if (d == 64) EV_GO(64, 128) else EV_GO(128, 128) }
#undef EV_GO
#undef EV_GO_STAGED
#undef EV_GO2
#undef EV_GO3
    SSLREC_LAUNCH_CHECK();
    const int n_cand = n_split * k;
#define EV_MERGE(PP) hipLaunchKernelGGL(eval_topk_merge_kernel<PP>, dim3((n_users + 3) / 4), dim3(256), 0, st, part_key, n_users, n_cand, k, out_idx, out_val)
    if (n_cand <= 64) EV_MERGE(1); else if (n_cand <= 128) EV_MERGE(2); else if (n_cand <= 256) EV_MERGE(4);
    else if (n_cand <= 512) EV_MERGE(8); else if (n_cand <= 1024) EV_MERGE(16); else if (n_cand <= 2048) EV_MERGE(32); else EV_MERGE(64);
#undef EV_MERGE
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---- full_predict: the dense [B, I] score matrix the reference's API returns ------------------------------------------
// `full_predict` + `_mask_predict` (reference models/general_cf/lightgcn.py:58-66, models/base_model.py:35-36):
//   out[u, i] = s * (1 - m) - 1e8 * m,   s = <UE[users[u]], IE[i]>,   m = train_mask[u, i]   (an int64 tensor in the reference:
// trainer/trainer.py moves the dataloader's dense mask with .long()), in ONE pass: exact-fp32 MFMA score tiles (a wave keeps 32 users
// resident and streams the item table), the mask read and the result written once, both in 128- / 256-byte runs -- the reference runs a GEMM and three elementwise passes over the 375 MB matrix (1024
// users at amazon-book size).  The in-tree Metric never needs the matrix (sslrec_eval_topk_f32); this entry point exists because
// full_predict's dense return value IS the reference's plugin contract.
template <int D, typename MaskT>
__global__ __launch_bounds__(256, 2) void full_predict_kernel(const float *__restrict__ UE, const int64_t *__restrict__ users, int n_users,
                                                           const float *__restrict__ IE, int n_items, const MaskT *__restrict__ mask,
                                                           int n_ugroup, int items_per_split, float *__restrict__ out) {
    constexpr int HALF = D / 2;
    const int lane = threadIdx.x & 63, h = lane >> 5, ur = lane & 31, w = wave_in_block();
    const int ug = blockIdx.x % n_ugroup, split = blockIdx.x / n_ugroup;
    const int u0 = (ug * 4 + w) * 32;
    if (u0 >= n_users) return;
    const int upos = min(u0 + ur, n_users - 1);
    const int64_t uid = users ? users[upos] : (int64_t)upos;
    float e1[HALF];
    ev_load_frag<D>(e1, UE, uid, lane);
    const int j_begin = split * items_per_split;
    const int j_end = min(j_begin + items_per_split, n_items);
    float fa[HALF], fb[HALF];
    // Scores NOT transposed here (users are the A operand): C register r of lane (c, h) is user u0 + crow(r, h), item j0 + c -- the 32
    // lanes of a half-wave hold 32 CONSECUTIVE items of one user, so a store instruction writes two 128-byte runs and a mask load reads
    // two 256-byte runs (the top-k kernel wants a user's items in one lane instead; written that way this kernel took 1.66 ms for
    // 1024 users at amazon-book size against 1.23 ms for the stock expression: 64 scattered dwords per store instruction)
    auto tile = [&](const float (&cur_frag)[HALF], float (&next_frag)[HALF], const int j0) {
        ev_load_frag<D>(next_frag, IE, min(j0 + 32 + ur, n_items - 1), lane);      // (unconditional: past the end the last row once more)
        ev_f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(e1[kk], cur_frag[kk], s, 0, 0, 0);      // s[user][item]
        const int j = j0 + ur;
        // All 16 mask elements are requested (as stored: no conversion yet) before the first is used -- rows and the column clamped into
        // the matrix, so that the loads are unconditional.  With a load inside every guarded store each store waited for its own mask
        // word (`s_waitcnt vmcnt(0)` sixteen times per tile): 0.60 ms for 1024 users at amazon-book size, 0.43 ms this way.
        if (mask) {
            MaskT raw[16];
            const size_t jc = (size_t)min(j, n_items - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) raw[r] = mask[(size_t)min(u0 + ev_crow(r, h), n_users - 1) * (size_t)n_items + jc];
            if (j >= j_end) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + ev_crow(r, h);
                const float m = (float)raw[r];
                if (u < n_users) out[(size_t)u * (size_t)n_items + j] = s[r] * (1.f - m) - 1e8f * m;
            }
        } else {
            if (j >= j_end) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + ev_crow(r, h);
                if (u < n_users) out[(size_t)u * (size_t)n_items + j] = s[r];
            }
        }
    };
    if (j_begin < j_end) ev_load_frag<D>(fa, IE, min(j_begin + ur, n_items - 1), lane);
    for (int j0 = j_begin; j0 < j_end; j0 += 64) {
        tile(fa, fb, j0);
        if (j0 + 32 < j_end) tile(fb, fa, j0 + 32);
    }
}

extern "C" int sslrec_full_predict_f32(const float *UE, const int64_t *users, int32_t n_users, const float *IE, int32_t n_items, int32_t d,
                                       const void *train_mask, int32_t mask_elem_bytes, float *out, void *stream) {
    if (!UE || !IE || !out || n_users <= 0 || n_items <= 0 || (d != 32 && d != 64 && d != 128)) return SSLREC_E_BADARG;
    if (train_mask && mask_elem_bytes != 8 && mask_elem_bytes != 4 && mask_elem_bytes != 1) return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int n_ugroup = (n_users + 127) / 128;
    const int tiles = (n_items + 31) / 32;
    int n_split = (1024 + n_ugroup - 1) / n_ugroup;           // ~4 workgroups per CU: the kernel streams, it does not wait on a chain
    if (n_split > tiles / 8) n_split = tiles / 8 > 1 ? tiles / 8 : 1;
    const int items_per_split = ((n_items + n_split - 1) / n_split + 31) / 32 * 32;
    n_split = (n_items + items_per_split - 1) / items_per_split;
#define FP_GO(DD, TT) hipLaunchKernelGGL((full_predict_kernel<DD, TT>), dim3(n_ugroup * n_split), dim3(256), 0, st, UE, users, n_users, IE, \
                                         n_items, (const TT *)train_mask, n_ugroup, items_per_split, out)
#define FP_BY_D(TT) { if (d == 32) FP_GO(32, TT); else if (d == 64) FP_GO(64, TT); else FP_GO(128, TT); }
    if (!train_mask || mask_elem_bytes == 8) FP_BY_D(int64_t)      // (the reference's mask: a .long() tensor)
    else if (mask_elem_bytes == 4) FP_BY_D(float)
    else FP_BY_D(uint8_t)
#undef FP_BY_D
#undef FP_GO
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---- negative sampler ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_negs_kernel(const int64_t *__restrict__ users, int64_t n, const int64_t *__restrict__ rowptr,
                                                          const int64_t *__restrict__ col, int32_t n_item,
                                                          const uint64_t *__restrict__ philox, uint32_t stream, int64_t *__restrict__ negs) {
    const PhiloxKey key = philox_load(philox);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t u = users[i];
        const int64_t lo = rowptr[u], hi = rowptr[u + 1];
        int64_t neg = 0;
        for (uint32_t attempt = 0; attempt < 4096; ++attempt) {      // a user with every item seen would never terminate
            // draw `attempt` of interaction i: counter (i, attempt) -- 4 candidates per Philox call
            const uint4 r = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32) ^ (attempt << 8), stream, key.step, key.k0, key.k1);
            const uint32_t c[4] = {r.x, r.y, r.z, r.w};
            bool done = false;
#pragma unroll
            for (int q = 0; q < 4 && !done; ++q) {
                neg = (int64_t)(((uint64_t)c[q] * (uint64_t)n_item) >> 32);          // uniform on [0, n_item)
                done = !ev_seen(col, lo, hi, neg);
            }
            if (done) break;
        }
        negs[i] = neg;
    }
}

extern "C" int sslrec_sample_negs(const int64_t *users, int64_t n, const int64_t *trn_rowptr, const int64_t *trn_col, int32_t n_item,
                                  const uint64_t *philox_state, uint32_t philox_stream, int64_t *negs_out, void *stream) {
    if (!users || n < 0 || !trn_rowptr || !trn_col || n_item <= 0 || !philox_state || !negs_out) return SSLREC_E_BADARG;
    if (n == 0) return 0;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(sample_negs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, users, n, trn_rowptr, trn_col, n_item,
                       philox_state, philox_stream, negs_out);
    SSLREC_LAUNCH_CHECK();
    return 0;
}
