// Host-side negative sampler that continues numpy's GLOBAL generator bit for bit.
//
// What it replaces: PairwiseTrnData.sample_negs (data_utils/datasets_general_cf.py:13-20) -- one Python iteration per train
// interaction: `np.random.randint(item_num)` until the (user, item) pair is not in the dok matrix (2.2 us per interaction
// measured, ~5 s per amazon-book epoch).  Here the same draws are taken from the same generator state in C++:
//
//   * numpy's legacy RandomState.randint(high) with high - 1 <= 0xFFFFFFFF takes ONE 32-bit output of MT19937 per attempt,
//     masks it with the smallest 2^k - 1 >= high - 1 and rejects values above high - 1 (numpy/random/_bounded_integers:
//     _rand_int64 -> random_bounded_uint64_fill -> buffered_bounded_masked_uint32, use_masked = True for RandomState);
//     high == 1 consumes nothing;
//   * MT19937 itself is the reference implementation (Matsumoto & Nishimura 1998; numpy/random/src/mt19937/mt19937.c):
//     624-word state, regenerated when pos == 624, tempering on output.
//
// The caller passes `np.random.get_state()`'s key / pos and writes them back with `np.random.set_state`, so whatever the
// reference draws next from numpy (the next epoch's negatives) continues from the same point.
#include <stdint.h>

#include "../../include/sslrec_hip.h"

namespace {

constexpr int kN = 624, kM = 397;

inline void mt_regenerate(uint32_t *mt) {
    auto mix = [](uint32_t hi, uint32_t lo) -> uint32_t {
        uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return (y >> 1) ^ ((lo & 1u) ? 0x9908b0dfu : 0u);
    };
    int k = 0;
    for (; k < kN - kM; ++k) mt[k] = mt[k + kM] ^ mix(mt[k], mt[k + 1]);
    for (; k < kN - 1; ++k) mt[k] = mt[k + (kM - kN)] ^ mix(mt[k], mt[k + 1]);
    mt[kN - 1] = mt[kM - 1] ^ mix(mt[kN - 1], mt[0]);
}

struct Mt {
    uint32_t *key;
    int pos;
    inline uint32_t next() {
        if (pos >= kN) {
            mt_regenerate(key);
            pos = 0;
        }
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

// is `item` one of row[0..len) (ascending)?
inline bool in_sorted_row(const int32_t *row, int64_t len, int32_t item) {
    int64_t lo = 0, hi = len;
    while (hi - lo > 8) {
        int64_t mid = (lo + hi) >> 1;
        if (row[mid] <= item) lo = mid; else hi = mid;
    }
    for (int64_t k = lo; k < hi; ++k)
        if (row[k] == item) return true;
    return false;
}

}  // namespace

extern "C" int sslrec_sample_negs_mt19937(uint32_t *mt_key, int32_t *mt_pos, const int32_t *users, int64_t n,
                                          const int64_t *trn_rowptr, const int32_t *trn_col, int32_t n_user,
                                          int32_t n_item, int32_t *negs_out, int64_t *n_draws) {
    if (!mt_key || !mt_pos || (n > 0 && (!users || !negs_out)) || !trn_rowptr || n_item < 1 || n < 0 || *mt_pos < 0 || *mt_pos > kN)
        return SSLREC_E_BADARG;
    const uint32_t rng = (uint32_t)(n_item - 1);
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t e = 0; e < n; ++e) {      // the reference would never return for a user who interacted with every item: refuse up front
        const int32_t u = users[e];
        if (u < 0 || u >= n_user) return SSLREC_E_BADARG;
        if (trn_rowptr[u + 1] - trn_rowptr[u] >= (int64_t)n_item) return SSLREC_E_BADARG;
    }
    Mt g{mt_key, *mt_pos};
    int64_t draws = 0;
    for (int64_t e = 0; e < n; ++e) {
        const int32_t u = users[e];
        const int32_t *row = trn_col + trn_rowptr[u];
        const int64_t len = trn_rowptr[u + 1] - trn_rowptr[u];
        int32_t cand;
        for (;;) {
            if (rng == 0) {
                cand = 0;                  // randint(1): no generator output is consumed
            } else {
                uint32_t v;
                do { v = g.next() & mask; ++draws; } while (v > rng);
                cand = (int32_t)v;
            }
            if (!in_sorted_row(row, len, cand)) break;
        }
        negs_out[e] = cand;
    }
    *mt_pos = g.pos;
    if (n_draws) *n_draws = draws;
    return 0;
}
