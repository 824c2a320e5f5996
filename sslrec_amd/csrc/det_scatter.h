// Deterministic scatter-add of gradient rows (shared by losses.hip and infonce.hip).
#pragma once
#include "common.h"

// ---- deterministic scatter-add of gradient rows ---------------------------------------------------------------------
// The backward of a gather is a scatter-add; the reference's index_put accumulates duplicates in an unspecified order,
// and float atomics would make two runs of the same step differ in the last bit.  Here every contribution e (K <= DET_MAX
// of them, staged as rows G[e]) registers its destination row in a small open-addressing hash table with INTEGER atomics
// whose results do not depend on the order of arrival: the slot's contributor count and its smallest contributor id.
// Then one wave per contribution: a destination with a single contributor (almost all) is updated directly; for a
// duplicated destination the smallest contributor adds ALL its contributions in ascending id order (ids kept in a short
// per-slot list and sorted, or found by scanning when the list overflowed) and writes the row once.  Bit-reproducible,
// no float atomics.  K > DET_MAX (batches beyond 5,461 triples) falls back to atomic adds.
#define DET_MAX 16384
#define DET_SLOTS 32768          // power of two, >= 2 * DET_MAX
#define DET_LIST 8

struct DetTable {                // lives in the caller's workspace
    unsigned long long *key;     // [DET_SLOTS] destination row address, 0 = empty
    int *cnt, *first;            // [DET_SLOTS] contributors / smallest contributor id
    int *list;                   // [DET_SLOTS][DET_LIST] contributor ids in arrival order (first DET_LIST of them)
    int *slot_of;                // [K] slot of contribution e
};

__host__ __device__ inline size_t det_ws_bytes(int K) {
    return (size_t)DET_SLOTS * (8 + 4 + 4 + 4 * DET_LIST) + (size_t)K * 4 + 64;
}

static DetTable det_table(void *ws) {
    DetTable t;
    char *p = (char *)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
    t.key = (unsigned long long *)p; p += (size_t)DET_SLOTS * 8;
    t.cnt = (int *)p; p += (size_t)DET_SLOTS * 4;
    t.first = (int *)p; p += (size_t)DET_SLOTS * 4;
    t.list = (int *)p; p += (size_t)DET_SLOTS * 4 * DET_LIST;
    t.slot_of = (int *)p;
    return t;
}

static __global__ __launch_bounds__(256) void det_clear_kernel(DetTable t) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < DET_SLOTS; i += gridDim.x * 256) {
        t.key[i] = 0ull;
        t.cnt[i] = 0;
        t.first[i] = 0x7fffffff;
    }
}

// registers contribution e for destination row `row_ptr` (one lane per contribution)
__device__ __forceinline__ void det_insert(const DetTable &t, const float *row_ptr, int e) {
    const unsigned long long k = (unsigned long long)(uintptr_t)row_ptr;
    unsigned int slot = (unsigned int)((k >> 4) * 0x9E3779B97F4A7C15ull >> 40) & (DET_SLOTS - 1);
    while (true) {
        const unsigned long long old = atomicCAS(&t.key[slot], 0ull, k);
        if (old == 0ull || old == k) break;
        slot = (slot + 1) & (DET_SLOTS - 1);
    }
    const int pos = atomicAdd(&t.cnt[slot], 1);
    atomicMin(&t.first[slot], e);
    if (pos < DET_LIST) t.list[(size_t)slot * DET_LIST + pos] = e;
    t.slot_of[e] = slot;
}

// one wave per contribution e: dst_row(e) += sum of the contributions of its slot, in ascending id order.
// self_clean: the wave that writes a slot's row also returns the slot to its cleared state (key 0, cnt 0, first MAX), so the NEXT call
// on the same table needs no clearing launch.  The other contributors of a duplicated slot only ever read cnt / first to find out
// that they are not the owner: before the clear they see cnt >= 2 and first = the owner, after it cnt = 0 and first = MAX -- either
// way "not me".
static __global__ __launch_bounds__(256) void det_reduce_kernel(DetTable t, int K, const float *__restrict__ G, int d, int self_clean) {
    const int lane = threadIdx.x & 63;
#define DET_RELEASE_SLOT() if (self_clean && lane == 0) { t.key[slot] = 0ull; t.cnt[slot] = 0; t.first[slot] = 0x7fffffff; }
    for (int e = blockIdx.x * 4 + wave_in_block(); e < K; e += gridDim.x * 4) {
        const int slot = t.slot_of[e];
        if (slot < 0) continue;                               // contribution stored directly (un-indexed role)
        const int c = t.cnt[slot];
        float *row = reinterpret_cast<float *>((uintptr_t)t.key[slot]);
        if (c == 1) {
            for (int k = lane; k < d; k += 64) row[k] += G[(size_t)e * d + k];
            DET_RELEASE_SLOT()
            continue;
        }
        if (t.first[slot] != e) continue;                     // the smallest contributor does the whole row
        if (c <= DET_LIST) {
            int ids[DET_LIST];
#pragma unroll
            for (int i = 0; i < DET_LIST; ++i) ids[i] = i < c ? t.list[(size_t)slot * DET_LIST + i] : 0x7fffffff;
#pragma unroll
            for (int i = 1; i < DET_LIST; ++i)                // insertion sort of <= 8 ids (wave-uniform)
#pragma unroll
                for (int j = i; j > 0; --j)
                    if (ids[j] < ids[j - 1]) { const int x = ids[j]; ids[j] = ids[j - 1]; ids[j - 1] = x; }
            for (int k = lane; k < d; k += 64) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < DET_LIST; ++i)
                    if (i < c) acc += G[(size_t)ids[i] * d + k];
                row[k] += acc;
            }
            DET_RELEASE_SLOT()
        } else {                                              // a heavily duplicated row: scan all contributions in id order
            float acc[4] = {0.f, 0.f, 0.f, 0.f};                // d <= 256: up to 4 floats per lane
            for (int j0 = 0; j0 < K; j0 += 64) {
                const int j = j0 + lane;
                unsigned long long m = __ballot(j < K && t.slot_of[j] == slot);
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int id = j0 + b;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (lane + 64 * q < d) acc[q] += G[(size_t)id * d + lane + 64 * q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane + 64 * q < d) row[lane + 64 * q] += acc[q];
            DET_RELEASE_SLOT()
        }
    }
#undef DET_RELEASE_SLOT
}


// grid-stride clear usable from inside another kernel's prologue (the table must be clear before the first det_insert of a call)
__device__ __forceinline__ void det_clear_from(const DetTable &t, int tid, int n_threads) {
    for (int i = tid; i < DET_SLOTS; i += n_threads) {
        t.key[i] = 0ull;
        t.cnt[i] = 0;
        t.first[i] = 0x7fffffff;
    }
}

static inline int det_reduce(const DetTable &tab, int K, const float *G, int d, hipStream_t st, int self_clean = 0) {
    int blocks = (K + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(det_reduce_kernel, dim3(blocks), dim3(256), 0, st, tab, K, G, d, self_clean);
    SSLREC_LAUNCH_CHECK();
    return 0;
}
