// All-rank evaluation and negative sampling on the device.
//
// 1. sslrec_eval_topk_f32 replaces, per evaluation batch, `full_predict` + `_mask_predict` + `t.topk`
//    (reference models/general_cf/lightgcn.py:58-66, models/base_model.py:35-36, trainer/metrics.py:99-103): the
//    reference ships a dense [B, I] int64 train mask from the host (750 MB per batch of 1024 at amazon-book size),
//    materializes the [B, I] score matrix, rewrites it three times and sorts it.  Here a wave keeps 32 users resident
//    as the B operand of exact-fp32 MFMA tiles (v_mfma_f32_32x32x2_f32, scores TRANSPOSED so that one lane sees 16
//    items of ONE user per tile, as in infonce.hip), streams the item table once, and keeps a k-entry candidate list
//    per lane in LDS.  A lane meets its user's items in ascending order, so the train mask is a MERGE: a per-lane
//    cursor into the user's sorted train row tells whether the next item is a seen one (seen items are skipped -- in
//    the reference they get -1e8 and lose to every unseen item); a score only costs more than two compares when it
//    beats the lane's current k-th best (~k ln(I/k) times per user).  A second small kernel merges the per-split lists into the final top-k,
//    descending by score, ties by ascending item id.  Nothing of size B x I ever exists.
// 2. sslrec_sample_negs replaces PairwiseTrnData.sample_negs (data_utils/datasets_general_cf.py:13-20: one
//    `np.random.randint(item_num)` per interaction, redrawn while the pair is a train interaction -- a Python loop
//    with dok lookups, 2.2 us per edge): one lane per interaction, Philox draws (philox.h), binary search in the
//    user's sorted train row.  Same distribution, different random stream.
#include "common.h"
#include "philox.h"

typedef float ev_f32x4 __attribute__((ext_vector_type(4)));
typedef float ev_f32x16 __attribute__((ext_vector_type(16)));

#define EVAL_KMAX 64

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

// partial lists: part_val / part_idx [n_users][n_split][2][k]
template <int D>
__global__ __launch_bounds__(256) void eval_topk_kernel(const float *__restrict__ UE, const int64_t *__restrict__ users, int n_users,
                                                        const float *__restrict__ IE, int n_items,
                                                        const int64_t *__restrict__ trn_rowptr, const int64_t *__restrict__ trn_col,
                                                        int k, int n_ugroup, int items_per_split, int n_split,
                                                        float *__restrict__ part_val, int32_t *__restrict__ part_idx) {
    extern __shared__ float lds[];                    // [4 waves][k][64 lanes] values, then the same shape of item ids
    constexpr int HALF = D / 2;
    const int lane = threadIdx.x & 63, h = lane >> 5, w = wave_in_block();
    const int ug = blockIdx.x % n_ugroup, split = blockIdx.x / n_ugroup;
    const int u0 = (ug * 4 + w) * 32;
    if (u0 >= n_users) return;
    float *cv = lds + (size_t)w * k * 64 + lane;                                        // slot s at cv[s * 64]
    int32_t *ci = reinterpret_cast<int32_t *>(lds + (size_t)4 * k * 64) + (size_t)w * k * 64 + lane;
    for (int s = 0; s < k; ++s) { cv[s * 64] = -INFINITY; ci[s * 64] = -1; }
    const int upos = min(u0 + (lane & 31), n_users - 1);
    const int64_t uid = users ? users[upos] : (int64_t)upos;
    float e1[HALF];
    ev_load_frag<D>(e1, UE, uid, lane);
    const int64_t row_hi = trn_rowptr ? trn_rowptr[uid + 1] : 0;
    float thr = -INFINITY;                               // the smallest value in this lane's list
    int thr_slot = 0;
    const int j_begin = split * items_per_split;
    const int j_end = min(j_begin + items_per_split, n_items);
    // cursor into the user's train row: first train item >= the first item this lane will see
    int64_t cur = trn_rowptr ? trn_rowptr[uid] : 0;
    {
        int64_t lo = cur, hi = row_hi;
        const int64_t first = j_begin + 4 * h;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (trn_col[mid] < first) lo = mid + 1; else hi = mid;
        }
        cur = lo;
    }
    int64_t next_seen = cur < row_hi ? trn_col[cur] : (int64_t)0x7fffffffffffffffll;
    float an[HALF];
    if (j_begin < j_end) ev_load_frag<D>(an, IE, min(j_begin + (lane & 31), n_items - 1), lane);
    for (int j0 = j_begin; j0 < j_end; j0 += 32) {
        float nx[HALF];
        const bool more = j0 + 32 < j_end;
        if (more) ev_load_frag<D>(nx, IE, min(j0 + 32 + (lane & 31), n_items - 1), lane);
        ev_f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(an[kk], e1[kk], s, 0, 0, 0);      // s[item][user]
        float best = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {                   // items in ascending order: merge against the sorted train row
            const int item = j0 + ev_crow(r, h);
            while (next_seen < item) { ++cur; next_seen = cur < row_hi ? trn_col[cur] : (int64_t)0x7fffffffffffffffll; }
            if (item >= j_end || next_seen == item) s[r] = -INFINITY;       // beyond the split, or a train item
            best = fmaxf(best, s[r]);
        }
        if (__ballot(best > thr)) {                      // rare after the first few hundred items
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int item = j0 + ev_crow(r, h);
                if (s[r] > thr) {
                    cv[thr_slot * 64] = s[r];
                    ci[thr_slot * 64] = item;
                    thr = INFINITY;                          // new minimum of the list
                    for (int t = 0; t < k; ++t) {
                        const float v = cv[t * 64];
                        if (v < thr) { thr = v; thr_slot = t; }
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < HALF; ++q) an[q] = nx[q];
        }
    }
    if (u0 + (lane & 31) < n_users) {
        const size_t o = (((size_t)(u0 + (lane & 31)) * n_split + split) * 2 + h) * k;
        for (int s = 0; s < k; ++s) { part_val[o + s] = cv[s * 64]; part_idx[o + s] = ci[s * 64]; }
    }
}

// one wave per user: the k best of its n_cand candidates, descending by value, ties by ascending item id; missing
// candidates (fewer than k unseen items) come out as -1
__global__ __launch_bounds__(256) void eval_topk_merge_kernel(const float *__restrict__ part_val, const int32_t *__restrict__ part_idx,
                                                              int n_users, int n_cand, int k, int64_t *__restrict__ out_idx,
                                                              float *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + wave_in_block();
    if (u >= n_users) return;
    constexpr int PER = 64;                             // candidates per lane, at most (n_cand <= 4096)
    float v[PER];
    int id[PER];
    const int per = (n_cand + 63) / 64;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = i * 64 + lane;
        const bool ok = i < per && c < n_cand;
        v[i] = ok ? part_val[(size_t)u * n_cand + c] : -INFINITY;
        id[i] = ok ? part_idx[(size_t)u * n_cand + c] : -1;
        if (id[i] < 0) v[i] = -INFINITY;
    }
    for (int t = 0; t < k; ++t) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, bslot = -1;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (i < per && id[i] >= 0 && (v[i] > bv || (v[i] == bv && id[i] < bi))) { bv = v[i]; bi = id[i]; bslot = i; }
        }
        float wv = bv;
        int wi = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(wv, o, 64);
            const int oi = __shfl_xor(wi, o, 64);
            if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
        }
        const bool none = (wi == 0x7fffffff);
        if (lane == 0) {
            out_idx[(size_t)u * k + t] = none ? -1 : (int64_t)wi;
            if (out_val) out_val[(size_t)u * k + t] = none ? -INFINITY : wv;
        }
        if (!none && bslot >= 0 && bi == wi && bv == wv) {     // the owner retires the winner
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (i == bslot) { id[i] = -1; v[i] = -INFINITY; }
        }
    }
}

extern "C" size_t sslrec_eval_topk_ws_bytes(int32_t n_users, int32_t n_items, int32_t k) {
    if (n_users <= 0 || n_items <= 0 || k <= 0 || k > EVAL_KMAX) return 0;
    const int n_ugroup = (n_users + 127) / 128;
    int n_split = (768 + n_ugroup - 1) / n_ugroup;
    n_split = n_split < 1 ? 1 : (n_split > 32 ? 32 : n_split);
    return (size_t)n_users * n_split * 2 * k * 8;
}

extern "C" int sslrec_eval_topk_f32(const float *UE, const int64_t *users, int32_t n_users, const float *IE, int32_t n_items,
                                    int32_t d, const int64_t *trn_rowptr, const int64_t *trn_col, int32_t k, void *ws,
                                    int64_t *out_idx, float *out_val, void *stream) {
    if (!UE || !IE || n_users <= 0 || n_items <= 0 || k <= 0 || k > EVAL_KMAX || !ws || !out_idx || (d != 32 && d != 64 && d != 128) ||
        ((trn_rowptr == nullptr) != (trn_col == nullptr)))
        return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int n_ugroup = (n_users + 127) / 128;
    int n_split = (768 + n_ugroup - 1) / n_ugroup;
    n_split = n_split < 1 ? 1 : (n_split > 32 ? 32 : n_split);
    const int items_per_split = ((n_items + n_split - 1) / n_split + 31) / 32 * 32;
    float *part_val = (float *)ws;
    int32_t *part_idx = (int32_t *)(part_val + (size_t)n_users * n_split * 2 * k);
    const size_t lds = (size_t)4 * k * 64 * 8;
#define EV_GO(DD)                                                                                                         \
    {                                                                                                                     \
        hipError_t e = hipFuncSetAttribute((const void *)eval_topk_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL(eval_topk_kernel<DD>, dim3(n_ugroup * n_split), dim3(256), lds, st, UE, users, n_users, IE, n_items, \
                           trn_rowptr, trn_col, k, n_ugroup, items_per_split, n_split, part_val, part_idx);               \
    }
    if (d == 32) EV_GO(32) else if (d == 64) EV_GO(64) else EV_GO(128)
#undef EV_GO
    SSLREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(eval_topk_merge_kernel, dim3((n_users + 3) / 4), dim3(256), 0, st, part_val, part_idx, n_users, n_split * 2 * k, k,
                       out_idx, out_val);
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
