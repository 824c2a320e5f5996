// Column-swept SpMM with the OUTPUT rows accumulated in LDS (gfx950: 160 KiB per CU).
//
// Same operator as spmm.hip (reference models/general_cf/lightgcn.py:28-29 `t.spmm(adj, embeds)`, the
// layer SUM of lightgcn.py:41 and EmbedPerturb, aug_utils.py:125-132, fused into the flush), for the
// regime where the whole output table fits the chip's LDS: n_rows * d * 4 <= 256 CUs x ~157 KiB
// (amazon-book at d=64 is 36.9 MB of 40 MB).
//
// Why: the row-streamed kernel gathers X rows in an order set by the rows, so the L2 of an XCD (4 MiB)
// sees the 37 MB table as random traffic: 57 % of the gathers miss and hold a vector-L1 miss slot for
// ~600 clk instead of ~200.  Here one workgroup per CU owns a fixed set of output rows as LDS
// accumulators, and each of its lane groups walks ITS edges sorted by COLUMN.  All workgroups start
// together and their streams have equal length, so the 32 CUs of an XCD sweep the X table front to
// back in step and an X row is pulled through the fabric about once per XCD instead of once per miss
// (measured: 433-544 MB of fabric reads per launch against 790 MB, 97 us against 119 us).  An edge-dropped view is the
// same layout with every lane group's stream compacted (swept_compact_kernel) and shorter step counts.
//
// Layout (built by sslrec_amd/csrc/plan.cpp): block b owns `slots` (a row, or one interleaved chunk of a heavy row); its 16
// waves x G lane groups (G = 256/d rows per 16-byte-per-lane instruction) own disjoint slot sets, so no two lane groups ever
// update the same accumulator -> plain LDS read-modify-write, no atomics, deterministic.  On a bipartite adjacency the two
// row classes live on different XCDs (workgroup b runs on XCD b % 8), so an XCD's L2 sweeps one embedding table: fabric
// reads 440 -> 200 MB per launch on the amazon-book-shaped graph, L2 hit rate 57 -> 77 %.
// Stream metadata costs vector-memory issue slots like the gathers do (two 16-byte loads per 4 steps were a third of the
// kernel's VMEM instructions), so a wave's stream is stored in 64-dword blocks of S steps (S = 16, or 8 at d = 32) that ONE
// coalesced dword load per array fetches: the lane group's entry of step j sits in lane j of each of its 16-lane rows and is
// broadcast with a DPP row_newbcast.  A packed word is column | slot << 20 (-1 = pad).  The flush adds the chunks of a row
// in slot order, applies the epilogue and writes each output row once.
#include "common.h"
#include "philox.h"
#include "swept_fmt.h"
#include <stdlib.h>
#include <type_traits>

struct SweptArgs {
    const int32_t *pack;
    const float *val;
    const int32_t *w_start, *w_steps;
    const int32_t *wf_ptr, *cf_ptr, *frow, *fstart, *fn;
    int32_t n_slots;
    const float *X;
    // up to SSLREC_MAX_VIEWS epilogues of the SAME product (SimGCL's first layer: one A.E0, three views)
    int32_t n_views;
    float eps;
    float *Y[SSLREC_MAX_VIEWS];
    const float *noise[SSLREC_MAX_VIEWS];
    const float *acc_in[SSLREC_MAX_VIEWS];
    float *acc_out[SSLREC_MAX_VIEWS];
    const uint64_t *philox;                  // device-side noise: computed, not read (philox.h)
    uint32_t philox_stream[SSLREC_MAX_VIEWS];
    int32_t philox_noise[SSLREC_MAX_VIEWS];
    // embedding-column passes (PASSES kernels): the tables have row_stride4 float4 per row, this launch works on the
    // D / 4 float4 starting at col_off4
    int32_t row_stride4, col_off4, n_pass;
    const float *noise_sumsq;      // full-row noise norms (one-view launches on a column slice), nullable
    int32_t noise_rs4, noise_co4;  // Philox noise: float4 per FULL noise row / this launch's column offset inside it (0 / 0: the table's own)
    const uint32_t *x_bits;        // hint, nullable: bit r clear = row r of X is all zeros (its entries are skipped like pads: no accumulate, the
                                   // gather reads row 0 from the L1); the first backward product of a BPR gradient touches <= 3B of the rows
    int32_t n_sum_in;              // deferred layer sum (one-view launches): acc_out = ((acc_in + sum_in[0]) + ...) + y
    const float *sum_in[SSLREC_MAX_SUM_IN];
    const float *axpy_x;           // acc_out += axpy_alpha * (*axpy_scale or 1) * axpy_x (one-view launches)
    float axpy_alpha;
    const float *axpy_scale;
    unsigned long long *trace;     // diagnostic (sslrec_debug_swept_trace): wall clock at the start of every block of every wave
    unsigned long long *stamp;     // measurement hook (sslrec_debug_stamp_next_launch): launch duration by the device's wall clock
    int32_t nt_stores;             // how the flush writes its rows (SSLREC_SWEPT_NT_STORES): 2 (default) = write-through (sc1) stores -- the rows leave the
                                   // XCD's L2 as they are written instead of piling up dirty until the end of the kernel (measured: -2 us per launch);
                                   // 0 = plain stores, 1 = non-temporal stores (measured: +3 us)
    int32_t prio_mode;             // issue priority of the 4 waves of a SIMD (SSLREC_SWEPT_PRIO): 2 (default) = rotates per metadata block, 0 = off
                                   // (static-by-age and per-4-steps variants were measured and removed: EXPERIMENTS.md B 4.1b, profiles/r03)
    int32_t late_flush;            // experiment switch (SSLREC_SWEPT_LATE_FLUSH=1): every wave waits for the workgroup before it writes its rows
    // factorized normalization (sslrec_epilogue_t.row_scale / scale_flags): A = diag(r) P diag(c) with a 0/1 pattern P.  PATTERN launches
    // never read the value stream: they add the gathered rows as they are and the flush multiplies the row sum by r[row]
    const float *row_scale;
    int32_t scale_flags;
    int32_t acc_init;              // experiment switch (SSLREC_SWEPT_ACC_INIT=1): launches that write only acc_out start their one-slot rows' accumulators
                                   // from acc_in instead of zero (the read moves from the flush to the start of the kernel)
};
#define SWEPT_TRACE_MAXB 32

#define SWEPT_WAVES 16

typedef float sw_f32x4 __attribute__((ext_vector_type(4)));

// experiment: write-through store (sc1: the line leaves the XCD's L2 at once instead of waiting for the end-of-kernel write-back)
__device__ __forceinline__ void sw_store_sc1(sw_f32x4 *p, sw_f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// WPE = waves per SIMD the kernel is compiled for: 4 (one 1024-thread workgroup per CU, 128 VGPRs) or 8 (the half-size
// layout of small matrices: two workgroups per CU, 64 VGPRs)
// PAT = pattern product (SSLREC_SCALE_PATTERN): every entry has the weight 1, the value stream is not read
template <int D, bool PASSES, int WPE, bool PAT>
__global__ __launch_bounds__(1024, WPE) void spmm_swept_kernel(SweptArgs a) {
    extern __shared__ float4 acc[];
    constexpr int G = 256 / D;        // output rows per wave instruction
    constexpr int LPG = 64 / G;       // lanes per row, one float4 each
    constexpr int RV = D / 4;         // float4 per row (== LPG)
    constexpr int S = SweptFmt<D>::S;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = lane % LPG;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    stamp_begin(a.stamp);

    const int wid = blockIdx.x * SWEPT_WAVES + wave_in_block();
    const int nblk = a.w_steps[wid] / S;
    const int32_t *pl = a.pack + a.w_start[wid] + lane;
    const float *vl = a.val + a.w_start[wid] + lane;
    const int RS = PASSES ? a.row_stride4 : RV;                       // float4 per table row
    const int CO = PASSES ? a.col_off4 : 0;
    const char *__restrict__ Xb = reinterpret_cast<const char *>(a.X) + (size_t)CO * 16;
    // Everything the kernel needs from MEMORY before its first gather is requested here, and the accumulators are zeroed
    // while those requests are in flight (the barrier behind the zeroing orders LDS only).
    // the stream's first metadata block first, so that the first gathers do not wait for the flush records
    // (PAT is a compile-time constant: a pattern launch carries no load of the value stream at all)
#define SW_VAL(OFF) (PAT ? 0.f : vl[OFF])
    int pv_first = -1;
    float vv_first = 0.f;
    if (nblk > 0) { pv_first = pl[0]; vv_first = SW_VAL(0); }
    // Flush records, kept in registers until the flush (nothing depends on them before, and the flush would otherwise start
    // with two dependent memory round trips).  Two kinds (plan.cpp): rows of ONE slot belong to a lane group of THIS wave and
    // are written out by it as soon as its own sweep ends -- no barrier, the stores overlap the slower waves' gathers; rows
    // cut into chunks are added up by the whole workgroup after the barrier (one prefetched pass, more through the loop).
    const int rl = lane / RV, rs = lane % RV;
    constexpr int RPW = 64 / RV;                       // rows a wave flushes per pass
    constexpr int RPP = 1024 / RV;                     // rows the workgroup flushes per pass
    // (the 64-register build of the half-size layout has no room to carry records through the sweep: PFW = 0, all through the loop)
    // (the column-pass builds carry the row stride, column offset and pass count on top: 8 prefetched passes keep them free of scratch)
    constexpr int FU = 3, PFW = (WPE == 4) ? (PASSES ? 8 : 12) : 0, PFA = PFW > 0 ? PFW : 1;
    const int wf0 = a.wf_ptr[wid], wf1 = a.wf_ptr[wid + 1];
    const int wpasses = (wf1 - wf0 + RPW - 1) / RPW;
    const bool pre_acc = a.n_views == 1 && a.acc_out[0] != nullptr;
    // (a record rides through the sweep as ONE register: row | first slot << 20, -1 = none -- rows < 2^20 by the layout's column limit,
    // slots <= 4094)
    int frp[PFA];
#define SW_FR_ROW(u) (frp[u] == -1 ? -1 : (frp[u] & 0xFFFFF))
#define SW_FR_S0(u) ((int)((unsigned)frp[u] >> 20))
    // (requested here with clamped indices -- 2 x 12 unconditional loads in one straight line -- and packed behind the zeroing of the
    // accumulators: a predicated load per record, packed at once, was twelve dependent round trips at the start of every launch)
    int fr_row[PFA], fr_s0[PFA];
#pragma unroll
    for (int u = 0; u < PFW; ++u) {
        const int ic = max(min(wf0 + u * RPW + rl, wf1 - 1), 0);
        fr_row[u] = a.frow[ic];
        fr_s0[u] = a.fstart[ic];
    }
    const int cf0 = a.cf_ptr[blockIdx.x], cf1 = a.cf_ptr[blockIdx.x + 1];
    const int cpasses = (cf1 - cf0 + RPP - 1) / RPP;
    const int crl = tid / RV;
    int crow = -1, cs0 = 0, cn = 0;      // (the 64-register build has no room to carry them through the sweep: fetched behind it)
    if constexpr (WPE == 4)
        if (cf0 + crl < cf1) { crow = a.frow[cf0 + crl]; cs0 = a.fstart[cf0 + crl]; cn = a.fn[cf0 + crl]; }
    // experiment (SSLREC_SWEPT_ACC_INIT): a launch that writes only acc_out = acc_in + y starts the accumulators of its one-slot rows from
    // acc_in (requested here, written behind the zeroing by the wave that owns the slot), so the flush of those rows reads nothing
#ifdef SSLREC_SWEPT_ACC_INIT
#pragma unroll
    for (int u = 0; u < PFW; ++u) frp[u] = (u < wpasses && wf0 + u * RPW + rl < wf1) ? (fr_row[u] | (fr_s0[u] << 20)) : -1;
#endif
#ifdef SSLREC_SWEPT_ACC_INIT      // (a build of its own, tools/build_variant.sh: the default build carries neither the registers nor the code)
    const bool ainit = !PAT && WPE == 4 && a.acc_init && pre_acc && !a.Y[0] && a.n_sum_in == 0 && !a.noise[0] && !a.philox_noise[0];
    float4 ai[PFA];
    if constexpr (WPE == 4 && !PAT) {
        if (ainit) {
#pragma unroll
            for (int u = 0; u < PFW; ++u) {
                ai[u] = zero4;
                if (frp[u] != -1) ai[u] = reinterpret_cast<const float4 *>(a.acc_in[0])[(size_t)SW_FR_ROW(u) * RS + CO + rs];
            }
        }
    }
#else
    constexpr bool ainit = false;
#endif
    for (int i = tid; i < a.n_slots * RV; i += 1024) acc[i] = zero4;
#ifdef SSLREC_SWEPT_FULL_FENCE
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifndef SSLREC_SWEPT_ACC_INIT
#pragma unroll
    for (int u = 0; u < PFW; ++u) frp[u] = (u < wpasses && wf0 + u * RPW + rl < wf1) ? (fr_row[u] | (fr_s0[u] << 20)) : -1;
#endif
#ifdef SSLREC_SWEPT_ACC_INIT
    if constexpr (WPE == 4 && !PAT) {
        if (ainit) {
#pragma unroll
            for (int u = 0; u < PFW; ++u)
                if (frp[u] != -1) acc[SW_FR_S0(u) * RV + rs] = ai[u];      // (this wave's own slots: its later LDS operations follow in order)
        }
    }
#endif
    // diagnostic: time stamp at the start of metadata block B (tools/spmm_trace.py)
#define SW_TRACE(B) \
    if (a.trace && (B) < SWEPT_TRACE_MAXB - 3 && lane == 0) a.trace[(size_t)wid * SWEPT_TRACE_MAXB + (B)] = wall_clock64();
#define SW_TRACE_AT(SLOT) if (a.trace && lane == 0) a.trace[(size_t)wid * SWEPT_TRACE_MAXB + (SLOT)] = wall_clock64();
    // Every gather is UNCONDITIONAL (a pad reads row 0 and accumulates nothing): a predicated load sits in its own basic block and
    // the compiler then waits for ALL outstanding loads (s_waitcnt vmcnt(0)) before the first accumulate of a group; straight-line
    // loads let it count them (vmcnt(4..8)) -- the plain product's launch went from 76 to 70.7 us with this alone (profiles/r03).
#define SW_GATHER(DST, PK) DST = *reinterpret_cast<const sw_f32x4 *>(Xb + (size_t)(((PK) == -1) ? 0 : ((PK) & 0xFFFFF)) * (PASSES ? RS * 16 : D * 4) + sub * 16);
#define SW_ACCUM(PK, VV, XX)                                               \
    if ((PK) != -1) {                                                      \
        const int s = (int)((unsigned)(PK) >> 20) * RV + sub;              \
        float4 t = acc[s];                                                 \
        if constexpr (PAT) {                                               \
            t.x += XX[0]; t.y += XX[1]; t.z += XX[2]; t.w += XX[3];        \
        } else {                                                           \
            t.x = fmaf(VV, XX[0], t.x); t.y = fmaf(VV, XX[1], t.y);        \
            t.z = fmaf(VV, XX[2], t.z); t.w = fmaf(VV, XX[3], t.w);        \
        }                                                                  \
        acc[s] = t;                                                        \
    }
    // the zero-row hint's launches: most entries are masked out, so their gathers are PREDICATED off (a masked lane group costs the
    // texture path nothing; an unconditional read of row 0 would still cost a full wave load: 61 against 80 us measured, the sweep
    // being bound by the vector-memory path either way) -- at the price of the compiler draining all loads before every accumulate
#define SW_GATHER_P(DST, PK) DST = sw_f32x4{0.f, 0.f, 0.f, 0.f}; \
    if ((PK) != -1) DST = *reinterpret_cast<const sw_f32x4 *>(Xb + (size_t)((PK) & 0xFFFFF) * (PASSES ? RS * 16 : D * 4) + sub * 16);
#define SW_G4P(PV, O, P)                                                                                              \
    { const int k0 = sw_bcast<D, O>(PV), k1 = sw_bcast<D, O + 1>(PV), k2 = sw_bcast<D, O + 2>(PV), k3 = sw_bcast<D, O + 3>(PV); \
      SW_GATHER_P(P##0, k0) SW_GATHER_P(P##1, k1) SW_GATHER_P(P##2, k2) SW_GATHER_P(P##3, k3) }
#define SW_G4(PV, O, P)                                                                                               \
    { const int k0 = sw_bcast<D, O>(PV), k1 = sw_bcast<D, O + 1>(PV), k2 = sw_bcast<D, O + 2>(PV), k3 = sw_bcast<D, O + 3>(PV); \
      SW_GATHER(P##0, k0) SW_GATHER(P##1, k1) SW_GATHER(P##2, k2) SW_GATHER(P##3, k3) }
#define SW_A4(PV, VV, O, P)                                                                                            \
    { const int k0 = sw_bcast<D, O>(PV), k1 = sw_bcast<D, O + 1>(PV), k2 = sw_bcast<D, O + 2>(PV), k3 = sw_bcast<D, O + 3>(PV); \
      const float u0 = __int_as_float(sw_bcast<D, O>(__float_as_int(VV))), u1 = __int_as_float(sw_bcast<D, O + 1>(__float_as_int(VV))), \
                  u2 = __int_as_float(sw_bcast<D, O + 2>(__float_as_int(VV))), u3 = __int_as_float(sw_bcast<D, O + 3>(__float_as_int(VV))); \
      SW_ACCUM(k0, u0, P##0) SW_ACCUM(k1, u1, P##1) SW_ACCUM(k2, u2, P##2) SW_ACCUM(k3, u3, P##3) }
    // narrow rows (D = 16 / 8: G = 16 / 32 rows per instruction, S = 4 / 2 steps per metadata block): one gather group per
    // block, the metadata of the next two blocks in flight
#define SW_GS(PV, P)                                                                            \
    { const int k0 = sw_bcast<D, 0>(PV), k1 = sw_bcast<D, 1>(PV);                               \
      SW_GATHER(P##0, k0) SW_GATHER(P##1, k1)                                                   \
      if constexpr (S == 4) { const int k2 = sw_bcast<D, 2>(PV), k3 = sw_bcast<D, 3>(PV);       \
                              SW_GATHER(P##2, k2) SW_GATHER(P##3, k3) } }
#define SW_AS(PV, VV, P)                                                                        \
    { const int k0 = sw_bcast<D, 0>(PV), k1 = sw_bcast<D, 1>(PV);                               \
      const float u0 = __int_as_float(sw_bcast<D, 0>(__float_as_int(VV))), u1 = __int_as_float(sw_bcast<D, 1>(__float_as_int(VV))); \
      SW_ACCUM(k0, u0, P##0) SW_ACCUM(k1, u1, P##1)                                             \
      if constexpr (S == 4) { const int k2 = sw_bcast<D, 2>(PV), k3 = sw_bcast<D, 3>(PV);       \
          const float u2 = __int_as_float(sw_bcast<D, 2>(__float_as_int(VV))), u3 = __int_as_float(sw_bcast<D, 3>(__float_as_int(VV))); \
          SW_ACCUM(k2, u2, P##2) SW_ACCUM(k3, u3, P##3) } }
    if constexpr (S <= 4) {
        if (nblk > 0) {
            sw_f32x4 x0, x1, x2, x3, y0, y1, y2, y3;
            int pv = pv_first, pn = -1;
            float vv = vv_first, vn = 0.f;
            if (nblk > 1) { pn = pl[64]; vn = SW_VAL(64); }
            SW_GS(pv, x)
            const int wq = wave_in_block() >> 2;
            for (int b = 0; b < nblk; b += 2) {
                if (a.prio_mode == 2 && (b & 6) == 0) {      // every 8 blocks (16-32 steps)
                    switch (((b >> 3) + wq) & 3) {
                        case 0: __builtin_amdgcn_s_setprio(0); break;
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        default: __builtin_amdgcn_s_setprio(3); break;
                    }
                }
                int p2 = -1, p3 = -1;
                float v2 = 0.f, v3 = 0.f;
                if (b + 2 < nblk) { p2 = pl[(size_t)(b + 2) * 64]; v2 = SW_VAL((size_t)(b + 2) * 64); }
                if (b + 3 < nblk) { p3 = pl[(size_t)(b + 3) * 64]; v3 = SW_VAL((size_t)(b + 3) * 64); }
                SW_GS(pn, y)
                SW_AS(pv, vv, x)
                SW_GS(p2, x)
                SW_AS(pn, vn, y)
                pv = p2; vv = v2; pn = p3; vn = v3;
            }
        }
    } else if (nblk > 0) {
        sw_f32x4 x0, x1, x2, x3, y0, y1, y2, y3;
        int pv = pv_first;
        float vv = vv_first;
        // The 4 waves of a SIMD do identical work and the issue arbiter prefers the OLDEST at equal priority: measured (tools/
        // spmm_trace.py, profiles/r03), the youngest wave of every SIMD ended its sweep 10 us after the oldest (54 / 56 / 59 / 64 us
        // by age) and the workgroup's flush waited for it.  Rotating the priorities per metadata block gives every wave the same
        // share of every rank: all 16 waves end within 1.3 us of each other, the launch is 4-5 us shorter.
        const int wq = wave_in_block() >> 2;      // 0 = the oldest wave of its SIMD ... 3 = the youngest
        // Stream metadata is requested TWO blocks ahead (one block ahead measured the same): with the zero-row hint the block in between is
        // the time the bitmap word of every entry of the next block has to arrive (a lane holds one entry per block: one 4-byte load from a
        // table of n_rows / 8 bytes that lives in the L1).
        const uint32_t *__restrict__ xb = a.x_bits;
        const int last = nblk - 1;
#define SW_ZERO_ROW_TEST(PK, W) if (xb && (PK) != -1 && !(((W) >> ((PK) & 31)) & 1u)) PK = -1;
        if (xb) { const uint32_t w0 = pv == -1 ? 0u : xb[(pv & 0xFFFFF) >> 5]; SW_ZERO_ROW_TEST(pv, w0) }
        SW_G4(pv, 0, x)
        int pn = pl[(size_t)min(1, last) * 64];
        float vn = SW_VAL((size_t)min(1, last) * 64);
        for (int b = 0; b < nblk; ++b) {      // the next 4 gathers are always in flight while 4 steps accumulate
            SW_TRACE(b)
            if (a.prio_mode == 2) {
                switch ((b + wq) & 3) {
                    case 0: __builtin_amdgcn_s_setprio(0); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
            const size_t nb_off = (size_t)min(b + 2, last) * 64;      // unconditional (clamped): the last blocks re-read the last one
            const int pnn = pl[nb_off];
            const float vnn = SW_VAL(nb_off);
            uint32_t wn = 0u;
            if (xb) wn = xb[(pn == -1 ? 0 : (pn & 0xFFFFF)) >> 5];
            if (WPE == 4 && xb) {      // (the 64-register build has no room for both bodies: masked entries read row 0 there)
                SW_G4P(pv, 4, y)
                SW_A4(pv, vv, 0, x)
                if constexpr (S == 16) {
                    SW_G4P(pv, 8, x)
                    SW_A4(pv, vv, 4, y)
                    SW_G4P(pv, 12, y)
                    SW_A4(pv, vv, 8, x)
                    SW_ZERO_ROW_TEST(pn, wn)
                    SW_G4P(pn, 0, x)
                    SW_A4(pv, vv, 12, y)
                } else {
                    SW_ZERO_ROW_TEST(pn, wn)
                    SW_G4P(pn, 0, x)
                    SW_A4(pv, vv, 4, y)
                }
            } else {
                SW_G4(pv, 4, y)
                SW_A4(pv, vv, 0, x)
                if constexpr (S == 16) {
                    SW_G4(pv, 8, x)
                    SW_A4(pv, vv, 4, y)
                    SW_G4(pv, 12, y)
                    SW_A4(pv, vv, 8, x)
                    SW_G4(pn, 0, x)
                    SW_A4(pv, vv, 12, y)
                } else {
                    SW_G4(pn, 0, x)
                    SW_A4(pv, vv, 4, y)
                }
            }
            pv = pn; vv = vn;
            pn = pnn; vn = vnn;
        }
    }
    if (a.prio_mode) __builtin_amdgcn_s_setprio(0);
    SW_TRACE_AT(SWEPT_TRACE_MAXB - 3)          // this wave's sweep is over

    // flush: RV lanes per output row (aligned lane groups)
    // one output row (this lane's float4 of it): chunks added in slot order, epilogues, stores
    // the deferred layer sum's operands of one output row position: ((acc_in + sum_in[0]) + ...), the running sum's order
    auto sum_in_rows = [&](float4 sa, const size_t at) {
        float4 e[SSLREC_MAX_SUM_IN];
#pragma unroll
        for (int j = 0; j < SSLREC_MAX_SUM_IN; ++j)
            e[j] = j < a.n_sum_in ? reinterpret_cast<const float4 *>(a.sum_in[j])[at] : zero4;
#pragma unroll
        for (int j = 0; j < SSLREC_MAX_SUM_IN; ++j)
            if (j < a.n_sum_in) { sa.x += e[j].x; sa.y += e[j].y; sa.z += e[j].z; sa.w += e[j].w; }
        return sa;
    };
    // have_acc (a compile-time tag: the common, 12-fold unrolled call sites carry no code for the other case): acc_row already holds
    // the accumulator input of the row (prefetched), else it is read here
    // rsc = row_scale[row] (1 without it): a pattern launch's row sum times r[row] is the product's row; SSLREC_SCALE_Y / _ACC write
    // Y / acc_out times r[row] (the operand of the NEXT pattern launch: sslrec_epilogue_t.scale_flags); x * 1.f is exact
    auto flush_row = [&](const int row, const int s0, const int n, auto have_acc_tag, const float4 acc_row, const float rsc) {
        constexpr bool have_acc = decltype(have_acc_tag)::value;
        const bool live = row >= 0;
        float4 t = zero4;
        size_t at = 0;
        if (live) {
            t = acc[s0 * RV + rs];
            for (int k = 1; k < n; ++k) {
                const float4 w = acc[(s0 + k) * RV + rs];
                t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
            }
            at = (size_t)row * RS + CO + rs;
        }
        if constexpr (PAT) { t.x *= rsc; t.y *= rsc; t.z *= rsc; t.w *= rsc; }
        for (int k = 0; k < a.n_views; ++k) {
            float4 tk = t;
            if (a.noise[k] || a.philox_noise[k]) {      // y += eps * sign(y) * noise_row / max(|noise_row|, 1e-12); norm over the row's RV lanes
                float4 nz = zero4;
                // a computed draw is indexed in the FULL noise table (a column slice of a feature-sliced table: noise_rs4 / noise_co4)
                const size_t nat = a.noise_rs4 ? (size_t)(live ? row : 0) * a.noise_rs4 + a.noise_co4 + rs : at;
                if (live) nz = a.noise[k] ? reinterpret_cast<const float4 *>(a.noise[k])[at]
                                          : philox_uniform4(philox_load(a.philox), (uint64_t)nat, a.philox_stream[k]);
                float ss = nz.x * nz.x + nz.y * nz.y + nz.z * nz.z + nz.w * nz.w;
                if (a.noise_sumsq) {                 // the full row's norm is given (the launch holds a slice of its columns)
                    ss = live ? a.noise_sumsq[row] : 0.f;
                } else {
                    if constexpr (PASSES) {              // the other column blocks of the row belong to its norm
                        for (int p = 0; p < a.n_pass; ++p) {
                            if (p * RV == CO || !live) continue;
                            const size_t ap = (size_t)row * RS + p * RV + rs;
                            const float4 o4 = a.noise[k] ? reinterpret_cast<const float4 *>(a.noise[k])[ap]
                                                         : philox_uniform4(philox_load(a.philox), (uint64_t)ap, a.philox_stream[k]);
                            ss += o4.x * o4.x + o4.y * o4.y + o4.z * o4.z + o4.w * o4.w;
                        }
                    }
#pragma unroll
                    for (int o = RV / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                }
                const float nrm = fmaxf(sqrtf(ss), 1e-12f);
                tk.x = tk.x + ((nz.x / nrm) * sign_f(tk.x)) * a.eps;
                tk.y = tk.y + ((nz.y / nrm) * sign_f(tk.y)) * a.eps;
                tk.z = tk.z + ((nz.z / nrm) * sign_f(tk.z)) * a.eps;
                tk.w = tk.w + ((nz.w / nrm) * sign_f(tk.w)) * a.eps;
            }
            if (!live) continue;
            if (a.Y[k]) {
                const float ysc = (a.scale_flags & SSLREC_SCALE_Y) ? rsc : 1.f;
                const sw_f32x4 yv = sw_f32x4{tk.x * ysc, tk.y * ysc, tk.z * ysc, tk.w * ysc};
                if (a.nt_stores == 1) __builtin_nontemporal_store(yv, reinterpret_cast<sw_f32x4 *>(a.Y[k]) + at);
                else if (a.nt_stores == 2) sw_store_sc1(reinterpret_cast<sw_f32x4 *>(a.Y[k]) + at, yv);
                else reinterpret_cast<sw_f32x4 *>(a.Y[k])[at] = yv;
            }
            if (a.acc_out[k]) {
                float4 sa = acc_row;        // (an if, not a ?: -- the select would become a flat load through scratch)
                if constexpr (!have_acc) {
                    sa = reinterpret_cast<const float4 *>(a.acc_in[k])[at];
                    if constexpr (WPE == 4)      // (the 64-register build of the half-size layout does not take deferred sums: launch_swept_one)
                        if (a.n_sum_in) sa = sum_in_rows(sa, at);
                }
                sa.x += tk.x; sa.y += tk.y; sa.z += tk.z; sa.w += tk.w;
                if (a.axpy_x) {                      // + alpha * x row (the regularizer's gradient in the last backward product)
                    const float al = a.axpy_alpha * (a.axpy_scale ? *a.axpy_scale : 1.f);
                    const float4 xr = reinterpret_cast<const float4 *>(a.axpy_x)[at];
                    sa.x = fmaf(al, xr.x, sa.x); sa.y = fmaf(al, xr.y, sa.y); sa.z = fmaf(al, xr.z, sa.z); sa.w = fmaf(al, xr.w, sa.w);
                }
                const float asc = (a.scale_flags & SSLREC_SCALE_ACC) ? rsc : 1.f;
                sa.x *= asc; sa.y *= asc; sa.z *= asc; sa.w *= asc;
                if (a.nt_stores == 1) __builtin_nontemporal_store(sw_f32x4{sa.x, sa.y, sa.z, sa.w}, reinterpret_cast<sw_f32x4 *>(a.acc_out[k]) + at);
                else if (a.nt_stores == 2) sw_store_sc1(reinterpret_cast<sw_f32x4 *>(a.acc_out[k]) + at, sw_f32x4{sa.x, sa.y, sa.z, sa.w});
                else reinterpret_cast<float4 *>(a.acc_out[k])[at] = sa;
            }
        }
    };
#define SW_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if (a.late_flush) SW_LDS_BARRIER();         // the round-2 order, kept for A/B measurements on one box
    // (1) the rows this wave owns: their accumulator rows (`acc_in`, one view) are requested together, then only LDS reads of
    // the wave's own slots (LDS operations of one wave execute in order), adds and stores remain
    typedef std::integral_constant<bool, true> AccGiven;
    typedef std::integral_constant<bool, false> AccRead;
    const float *__restrict__ rsv = a.row_scale;
    // r[row] of the prefetched passes' rows: UNCONDITIONAL loads (a dead record reads r[0]) in one straight line under a wave-uniform
    // test -- as twelve predicated loads the compiler sank each one to its use inside flush_row: twelve dependent round trips, +4 us
    // per launch (profiles/r05/spmm_levers.json, call a)
    float rscp[PFA];
#pragma unroll
    for (int u = 0; u < PFW; ++u) rscp[u] = 1.f;
    if (rsv) {
#pragma unroll
        for (int u = 0; u < PFW; ++u) rscp[u] = rsv[frp[u] == -1 ? 0 : (frp[u] & 0xFFFFF)];
    }
    if (pre_acc && a.n_sum_in == 0) {
        float4 accp[PFA];
#pragma unroll
        for (int u = 0; u < PFW; ++u) {
            accp[u] = zero4;
            if (frp[u] != -1 && !ainit) accp[u] = reinterpret_cast<const float4 *>(a.acc_in[0])[(size_t)SW_FR_ROW(u) * RS + CO + rs];
        }
#pragma unroll
        for (int u = 0; u < PFW; ++u)
            if (u < wpasses) flush_row(SW_FR_ROW(u), SW_FR_S0(u), 1, AccGiven(), accp[u], rscp[u]);      // uniform condition
    } else if (pre_acc) {
        // deferred layer sum: 1 + n_sum_in table rows per output row -- CH passes' worth of them requested together
        constexpr int CH = 3;
#pragma unroll
        for (int u0 = 0; u0 < PFW; u0 += CH) {
            float4 base[CH], e[SSLREC_MAX_SUM_IN][CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int row = (u0 + c < PFW) ? SW_FR_ROW((u0 + c < PFW) ? u0 + c : 0) : -1;
                const size_t at = (size_t)(row >= 0 ? row : 0) * RS + CO + rs;
                base[c] = zero4;
                if (row >= 0) base[c] = reinterpret_cast<const float4 *>(a.acc_in[0])[at];
#pragma unroll
                for (int j = 0; j < SSLREC_MAX_SUM_IN; ++j) {
                    e[j][c] = zero4;
                    if (row >= 0 && j < a.n_sum_in) e[j][c] = reinterpret_cast<const float4 *>(a.sum_in[j])[at];
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
#pragma unroll
                for (int j = 0; j < SSLREC_MAX_SUM_IN; ++j)
                    if (j < a.n_sum_in) { base[c].x += e[j][c].x; base[c].y += e[j][c].y; base[c].z += e[j][c].z; base[c].w += e[j][c].w; }
                if (u0 + c < PFW && u0 + c < wpasses) flush_row(SW_FR_ROW((u0 + c < PFW) ? u0 + c : 0), SW_FR_S0((u0 + c < PFW) ? u0 + c : 0), 1, AccGiven(), base[c], rscp[(u0 + c < PFW) ? u0 + c : 0]);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < PFW; ++u)
            if (u < wpasses) flush_row(SW_FR_ROW(u), SW_FR_S0(u), 1, AccRead(), zero4, rscp[u]);      // uniform condition
    }
    for (int it0 = PFW; it0 < wpasses; it0 += FU) {      // waves with more rows than the prefetch covers: FU passes at a time
        int s0v[FU], rowv[FU];
        float rscv[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = wf0 + (it0 + u) * RPW + rl;
            const bool live = i < wf1;
            s0v[u] = live ? a.fstart[i] : 0;
            rowv[u] = live ? a.frow[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) rscv[u] = 1.f;
        if (rsv) {
#pragma unroll
            for (int u = 0; u < FU; ++u) rscv[u] = rsv[rowv[u] >= 0 ? rowv[u] : 0];
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (it0 + u < wpasses) flush_row(rowv[u], s0v[u], 1, AccRead(), zero4, rscv[u]);      // uniform condition
    }
    // (2) the chunked rows of the workgroup.  The barrier orders LDS only: __syncthreads() would also wait for the accumulator
    // rows requested just before it (vmcnt(0)), which is exactly the latency that request is meant to hide
    if constexpr (WPE != 4)
        if (cf0 + crl < cf1) { crow = a.frow[cf0 + crl]; cs0 = a.fstart[cf0 + crl]; cn = a.fn[cf0 + crl]; }
    float4 cacc = zero4;
    float crsc = 1.f;
    if (rsv) crsc = rsv[crow >= 0 ? crow : 0];
    if constexpr (WPE == 4) {
        if (pre_acc && crow >= 0) {
            const size_t at = (size_t)crow * RS + CO + rs;
            cacc = reinterpret_cast<const float4 *>(a.acc_in[0])[at];
            if (a.n_sum_in) cacc = sum_in_rows(cacc, at);
        }
    }
    if (!a.late_flush) SW_LDS_BARRIER();
    SW_TRACE_AT(SWEPT_TRACE_MAXB - 2)          // the workgroup's chunk flush starts
    if (cpasses > 0) {
        if (WPE == 4 && pre_acc) flush_row(crow, cs0, cn, AccGiven(), cacc, crsc);
        else flush_row(crow, cs0, cn, AccRead(), zero4, crsc);
    }
    for (int it = 1; it < cpasses; ++it) {
        const int i = cf0 + it * RPP + crl;
        const bool live = i < cf1;
        const int row_i = live ? a.frow[i] : -1;
        flush_row(row_i, live ? a.fstart[i] : 0, live ? a.fn[i] : 0, AccRead(), zero4, (rsv && row_i >= 0) ? rsv[row_i] : 1.f);
    }
    SW_TRACE_AT(SWEPT_TRACE_MAXB - 1)          // this wave's share of the flush is issued
    stamp_end<true>(a.stamp);
}

// Diagnostic (tools/spmm_trace.py), not part of the operator ABI: while enabled, every launch of the column-swept kernel
// records per wave the 100 MHz wall clock at the start of each metadata block (slots 0..28), at the end of its sweep (29),
// at the start (30) and at the end (31) of its flush, into a ring of SWEPT_TRACE_RING launches.
#define SWEPT_TRACE_RING 4
static unsigned long long *g_swept_trace = nullptr;
static size_t g_swept_trace_stride = 0;
static unsigned g_swept_trace_launch = 0;
extern "C" int sslrec_debug_swept_trace(int enable, unsigned long long *host_out, int n_waves) {
    const size_t stride = (size_t)n_waves * SWEPT_TRACE_MAXB;
    if (enable && !g_swept_trace) {
        if (hipMalloc((void **)&g_swept_trace, SWEPT_TRACE_RING * stride * 8) != hipSuccess) return SSLREC_E_BADARG;
        (void)hipMemset(g_swept_trace, 0, SWEPT_TRACE_RING * stride * 8);
        g_swept_trace_stride = stride;
        g_swept_trace_launch = 0;
    }
    if (host_out && g_swept_trace) {      // [SWEPT_TRACE_RING][n_waves][32]; launch k of the enabled period is in ring slot k % RING
        (void)hipDeviceSynchronize();
        if (hipMemcpy(host_out, g_swept_trace, SWEPT_TRACE_RING * g_swept_trace_stride * 8, hipMemcpyDeviceToHost) != hipSuccess) return SSLREC_E_BADARG;
    }
    if (!enable && g_swept_trace) { (void)hipFree(g_swept_trace); g_swept_trace = nullptr; }
    return (int)g_swept_trace_launch;
}

// in-kernel launch timing (include/sslrec_hip.h): the record attached to the next SpMM launch of the calling thread
static thread_local unsigned long long *g_next_stamp = nullptr;
extern "C" int sslrec_debug_stamp_next_launch(unsigned long long *record) {
    g_next_stamp = record;
    return 0;
}
extern "C" int sslrec_debug_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return khz;
}
unsigned long long *sslrec_take_stamp() {
    unsigned long long *r = g_next_stamp;
    g_next_stamp = nullptr;
    return r;
}

template <int D, bool PASSES, int WPE, bool PAT>
static int launch_swept_pat(const SweptArgs &a, int n_blocks, hipStream_t st) {
    const size_t lds = (size_t)a.n_slots * D * 4;
    static bool attr_set[64] = {};      // per instantiation and per device: the attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SSLREC_E_BADARG;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)spmm_swept_kernel<D, PASSES, WPE, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SSLREC_SWEPT_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    SweptArgs b = a;
    b.trace = nullptr;
    b.stamp = a.stamp;
    static const int late = [] { const char *e = getenv("SSLREC_SWEPT_LATE_FLUSH"); return (e && atoi(e) != 0) ? 1 : 0; }();
    b.late_flush = late;
    static const int prio = [] { const char *e = getenv("SSLREC_SWEPT_PRIO"); return e ? atoi(e) : 2; }();
    b.prio_mode = prio;
    static const int nts = [] { const char *e = getenv("SSLREC_SWEPT_NT_STORES"); return e ? atoi(e) : 2; }();
    b.nt_stores = nts;
    static const int ainit = [] { const char *e = getenv("SSLREC_SWEPT_ACC_INIT"); return (e && atoi(e) != 0) ? 1 : 0; }();
    b.acc_init = ainit;
    if (g_swept_trace && (size_t)n_blocks * SWEPT_WAVES * SWEPT_TRACE_MAXB <= g_swept_trace_stride)
        b.trace = g_swept_trace + (size_t)(g_swept_trace_launch++ % SWEPT_TRACE_RING) * g_swept_trace_stride;
    hipLaunchKernelGGL((spmm_swept_kernel<D, PASSES, WPE, PAT>), dim3(n_blocks), dim3(1024), lds, st, b);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

template <int D, bool PASSES, int WPE>
static int launch_swept_wpe(const SweptArgs &a, int n_blocks, hipStream_t st) {
    if (a.scale_flags & SSLREC_SCALE_PATTERN) return launch_swept_pat<D, PASSES, WPE, true>(a, n_blocks, st);
    return launch_swept_pat<D, PASSES, WPE, false>(a, n_blocks, st);
}

template <int D, bool PASSES>
static int launch_swept_one(const SweptArgs &a, int n_blocks, hipStream_t st) {
    // more workgroups than CUs (the builder's half-size layout): two must be resident per CU
    if (n_blocks > 256 || (size_t)a.n_slots * D * 4 <= SSLREC_SWEPT_LDS_BYTES / 2) {
        if (a.n_sum_in) return SSLREC_E_BADARG;      // the half-size layout's 64-register build keeps the running sum (sslrec_swept_deferred_sum_ok)
        return launch_swept_wpe<D, PASSES, 8>(a, n_blocks, st);
    }
    return launch_swept_wpe<D, PASSES, 4>(a, n_blocks, st);
}

// d_full == D: one launch; d_full = n_pass * D: one launch per block of D embedding columns
template <int D>
static int launch_swept(const SweptArgs &a0, int n_blocks, int d_full, hipStream_t st) {
    if (d_full == D) return launch_swept_one<D, false>(a0, n_blocks, st);
    SweptArgs a = a0;
    a.n_pass = d_full / D;
    a.row_stride4 = d_full / 4;
    for (int p = 0; p < a.n_pass; ++p) {
        a.col_off4 = p * (D / 4);
        const int rc = launch_swept_one<D, true>(a, n_blocks, st);
        if (rc != 0) return rc;
    }
    return 0;
}

// EdgeDrop on the swept layout (replaces EdgeDrop.forward, models/aug_utils.py:18-31): every lane group's stream is
// compacted in place of its own slots -- the kept entries keep their (column) order and move to the front, the tail
// becomes pads, and the wave's step count shrinks to the longest of its G compacted streams (whole metadata blocks).  One
// wave per stream wave; a lane group reads LPG consecutive steps of ITS stream per pass, a ballot gives every kept entry
// its rank; at d >= 128 an entry is stored once per 16-lane row of its lane group.
template <int D>
__global__ __launch_bounds__(256) void swept_compact_kernel(const int32_t *__restrict__ pack, const float *__restrict__ val,
                                                            const int32_t *__restrict__ w_start,
                                                            const int32_t *__restrict__ w_steps, int n_streams,
                                                            const int32_t *__restrict__ edge_map,
                                                            const uint8_t *__restrict__ keep, float keep_rate,
                                                            const uint64_t *__restrict__ philox, uint32_t philox_stream,
                                                            float scale, int32_t *__restrict__ pack_out,
                                                            float *__restrict__ val_out, int32_t *__restrict__ w_steps_out) {
    constexpr int G = 256 / D, LPG = 64 / G, S = SweptFmt<D>::S, COPIES = (LPG >= 16) ? LPG / 16 : 1;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block();
    if (w >= n_streams) return;
    const int sub = lane % LPG, g = lane / LPG;
    const int base = w_start[w] + g * LPG, steps = w_steps[w];
    // element of (step s, this lane group): block s / S, lane j = s % S of each of the group's 16-lane rows
#define SW_ELEM(s) (base + ((s) / S) * 64 + ((s) % S))
    const unsigned long long gmask = (LPG == 64) ? ~0ull : (((1ull << LPG) - 1ull) << (g * LPG));
    PhiloxKey pkey = {};
    if (!keep) pkey = philox_load(philox);            // the mask is computed: floor(u + keep_rate) != 0 (aug_utils.py:28-29)
    int count = 0;                                    // kept entries of this lane group so far (uniform in the group)
    for (int s0 = 0; s0 < steps; s0 += LPG) {
        const int s = s0 + sub;
        int pk = -1;
        float v = 0.f;
        bool kp = false;
        if (s < steps) {
            const int e = SW_ELEM(s);
            pk = pack[e];
            if (pk != -1) {
                const int k = edge_map[e];
                kp = keep ? keep[k] != 0 : floorf(philox_uniform1(pkey, (uint64_t)k, philox_stream) + keep_rate) != 0.f;
                v = val[e] * scale;
            }
        }
        const unsigned long long m = __ballot(kp) & gmask;
        if (kp) {
            const int so = count + __popcll(m & ((1ull << lane) - 1ull));
            const int o = SW_ELEM(so);
#pragma unroll
            for (int c = 0; c < COPIES; ++c) { pack_out[o + c * 16] = pk; val_out[o + c * 16] = v; }
        }
        count += __popcll(m);
    }
    int longest = count;                              // over the G lane groups of the wave
#pragma unroll
    for (int o = LPG; o < 64; o <<= 1) longest = max(longest, __shfl_xor(longest, o, 64));
    const int steps_out = (longest + S - 1) / S * S;
    for (int s = count + sub; s < steps_out; s += LPG) {
        const int o = SW_ELEM(s);
#pragma unroll
        for (int c = 0; c < COPIES; ++c) { pack_out[o + c * 16] = -1; val_out[o + c * 16] = 0.f; }
    }
#undef SW_ELEM
    if (lane == 0) w_steps_out[w] = steps_out;
}

static int swept_compact_any(const sslrec_swept_t *A, const int32_t *edge_map, const uint8_t *keep, float keep_rate,
                             const uint64_t *philox, uint32_t philox_stream, float scale, int32_t *pack_out, float *val_out,
                             int32_t *w_steps_out, void *stream) {
    if (!A || !edge_map || (!keep && !philox) || !pack_out || !val_out || !w_steps_out) return SSLREC_E_BADARG;
    const int n_streams = A->n_blocks * SWEPT_WAVES;
    const int blocks = (n_streams + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
#define SW_COMPACT(DD)                                                                                                  \
    hipLaunchKernelGGL(swept_compact_kernel<DD>, dim3(blocks), dim3(256), 0, st, A->pack, A->val, A->w_start, A->w_steps, \
                       n_streams, edge_map, keep, keep_rate, philox, philox_stream, scale, pack_out, val_out, w_steps_out)
    switch (A->d) {
        case 8: SW_COMPACT(8); break;
        case 16: SW_COMPACT(16); break;
        case 32: SW_COMPACT(32); break;
        case 64: SW_COMPACT(64); break;
        case 128: SW_COMPACT(128); break;
        case 256: SW_COMPACT(256); break;
        default: return SSLREC_E_BADARG;
    }
#undef SW_COMPACT
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_swept_compact(const sslrec_swept_t *A, const int32_t *edge_map, const uint8_t *keep, float scale,
                                    int32_t *pack_out, float *val_out, int32_t *w_steps_out, void *stream) {
    if (!keep) return SSLREC_E_BADARG;
    return swept_compact_any(A, edge_map, keep, 0.f, nullptr, 0, scale, pack_out, val_out, w_steps_out, stream);
}

extern "C" int sslrec_swept_compact_philox(const sslrec_swept_t *A, const int32_t *edge_map, float keep_rate,
                                           const uint64_t *philox_state, uint32_t philox_stream, float scale,
                                           int32_t *pack_out, float *val_out, int32_t *w_steps_out, void *stream) {
    if (!philox_state || !(keep_rate >= 0.f && keep_rate <= 1.f)) return SSLREC_E_BADARG;
    return swept_compact_any(A, edge_map, nullptr, keep_rate, philox_state, philox_stream, scale, pack_out, val_out, w_steps_out,
                             stream);
}

__global__ void philox_advance_kernel(uint64_t *state) { state[1] += 1; }

extern "C" int sslrec_philox_advance(uint64_t *philox_state, void *stream) {
    if (!philox_state) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(philox_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, philox_state);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void philox_fill_kernel(const uint64_t *state, uint32_t stream, float4 *out, size_t n4) {
    const PhiloxKey k = philox_load(state);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        out[i] = philox_uniform4(k, (uint64_t)i, stream);
}

extern "C" int sslrec_philox_fill_f32(const uint64_t *philox_state, uint32_t philox_stream, float *out, size_t n, void *stream) {
    if (!philox_state || !out || (n & 3)) return SSLREC_E_BADARG;
    const size_t n4 = n / 4;
    if (n4 == 0) return 0;
    const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipLaunchKernelGGL(philox_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, philox_state, philox_stream,
                       reinterpret_cast<float4 *>(out), n4);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// full-row norms of EmbedPerturb's noise for launches that hold a column slice of the row (include/sslrec_hip.h)
__global__ __launch_bounds__(256) void philox_row_sumsq_kernel(const uint64_t *state, uint32_t stream, int n_rows, int dv, float *out) {
    const PhiloxKey k = philox_load(state);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    float ss = 0.f;
    for (int c = 0; c < dv; ++c) {      // ascending column order
        const float4 u = philox_uniform4(k, (uint64_t)r * dv + c, stream);
        ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    }
    out[r] = ss;
}

extern "C" int sslrec_philox_row_sumsq(const uint64_t *philox_state, uint32_t philox_stream, int32_t n_rows, int32_t d, float *out,
                                       void *stream) {
    if (!philox_state || !out || n_rows < 0 || d <= 0 || (d & 3)) return SSLREC_E_BADARG;
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(philox_row_sumsq_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, philox_state, philox_stream,
                       n_rows, d / 4, out);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float4 *x, int n_rows, int dv, float *out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    float ss = 0.f;
    for (int c = 0; c < dv; ++c) {
        const float4 u = x[(size_t)r * dv + c];
        ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    }
    out[r] = ss;
}

extern "C" int sslrec_row_sumsq_f32(const float *x, int32_t n_rows, int32_t d, float *out, void *stream) {
    if (!x || !out || n_rows < 0 || d <= 0 || (d & 3) || ((uintptr_t)x & 15)) return SSLREC_E_BADARG;
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(row_sumsq_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), n_rows, d / 4, out);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

static int swept_dispatch(const SweptArgs &a, const sslrec_swept_t *A, int d, hipStream_t st) {
    switch (A->d) {
        case 8: return launch_swept<8>(a, A->n_blocks, d, st);
        case 16: return launch_swept<16>(a, A->n_blocks, d, st);
        case 32: return launch_swept<32>(a, A->n_blocks, d, st);
        case 64: return launch_swept<64>(a, A->n_blocks, d, st);
        case 128: return launch_swept<128>(a, A->n_blocks, d, st);
        case 256: return launch_swept<256>(a, A->n_blocks, d, st);
        default: return SSLREC_E_BADARG;
    }
}

// d = the tables' embedding size: the layout's own (A->d), or 2 / 4 / 8 times it (embedding-column passes)
static bool swept_ok(const sslrec_swept_t *A, const float *X, int d) {
    if (!A || !X || A->d <= 0 || d % A->d != 0 || A->n_blocks <= 0 || A->n_slots <= 0) return false;
    const int n_pass = d / A->d;
    if (n_pass != 1 && n_pass != 2 && n_pass != 4 && n_pass != 8) return false;
    return !((size_t)A->n_slots * A->d * 4 > SSLREC_SWEPT_LDS_BYTES || A->n_cols > (1 << 20) || A->n_slots > 4095);
}

// can launches on this layout take sslrec_epilogue_t.sum_in?  (the one-workgroup-per-CU build only)
extern "C" int sslrec_swept_deferred_sum_ok(const sslrec_swept_t *A) {
    if (!A || A->d <= 0 || A->n_blocks <= 0 || A->n_slots <= 0) return 0;
    return !(A->n_blocks > 256 || (size_t)A->n_slots * A->d * 4 <= SSLREC_SWEPT_LDS_BYTES / 2);
}

extern "C" int sslrec_spmm_swept_f32(const sslrec_swept_t *A, const int32_t *pack_override, const float *val_override,
                                     const int32_t *w_steps_override, const float *X, int32_t d, float *Y,
                                     const sslrec_epilogue_t *epi, void *stream) {
    if (!swept_ok(A, X, d)) return SSLREC_E_BADARG;
    if (!Y && !(epi && epi->acc_out)) return SSLREC_E_BADARG;
    if (epi && epi->acc_out && !epi->acc_in) return SSLREC_E_BADARG;
    SweptArgs a = {};
    a.pack = pack_override ? pack_override : A->pack;
    a.val = val_override ? val_override : A->val;
    a.w_start = A->w_start;
    a.w_steps = w_steps_override ? w_steps_override : A->w_steps;
    a.wf_ptr = A->wf_ptr; a.cf_ptr = A->cf_ptr; a.frow = A->f_row; a.fstart = A->f_start; a.fn = A->f_n;
    a.stamp = sslrec_take_stamp();
    a.n_slots = A->n_slots;
    a.X = X;
    a.n_views = 1;
    a.eps = epi ? epi->eps : 0.f;
    a.Y[0] = Y;
    a.noise[0] = epi ? epi->noise : nullptr;
    a.acc_in[0] = epi ? epi->acc_in : nullptr;
    a.acc_out[0] = epi ? epi->acc_out : nullptr;
    if (epi && !epi->noise && epi->philox) {
        a.philox = epi->philox;
        a.philox_stream[0] = epi->philox_stream;
        a.philox_noise[0] = 1;
    }
    if (epi) {
        if ((epi->noise_row_stride | epi->noise_col_off) & 3) return SSLREC_E_BADARG;
        if (epi->noise_row_stride && (d != A->d || !epi->noise_sumsq)) return SSLREC_E_BADARG;      // a slice runs in one pass and brings the row norms
        if (epi->axpy_x && !epi->acc_out) return SSLREC_E_BADARG;
        a.noise_sumsq = epi->noise_sumsq;
        a.noise_rs4 = epi->noise_row_stride / 4;
        a.noise_co4 = epi->noise_col_off / 4;
        a.axpy_x = epi->axpy_x; a.axpy_alpha = epi->axpy_alpha; a.axpy_scale = epi->axpy_scale;
        a.x_bits = epi->x_row_bits;
        if (epi->scale_flags & ~(SSLREC_SCALE_PATTERN | SSLREC_SCALE_Y | SSLREC_SCALE_ACC)) return SSLREC_E_BADARG;
        if (epi->scale_flags && !epi->row_scale) return SSLREC_E_BADARG;
        if ((epi->scale_flags & SSLREC_SCALE_PATTERN) && epi->x_row_bits) return SSLREC_E_BADARG;      // (the hinted launch is a valued one)
        a.row_scale = epi->scale_flags ? epi->row_scale : nullptr;
        a.scale_flags = epi->scale_flags;
        if (epi->n_sum_in < 0 || epi->n_sum_in > SSLREC_MAX_SUM_IN || (epi->n_sum_in && !epi->acc_out)) return SSLREC_E_BADARG;
        a.n_sum_in = epi->n_sum_in;
        for (int j = 0; j < epi->n_sum_in; ++j) {
            if (!epi->sum_in[j]) return SSLREC_E_BADARG;
            a.sum_in[j] = epi->sum_in[j];
        }
    }
    return swept_dispatch(a, A, d, (hipStream_t)stream);
}

extern "C" int sslrec_spmm_swept_views_f32(const sslrec_swept_t *A, const float *X, int32_t d,
                                           const sslrec_epilogue_views_t *views, void *stream) {
    if (!swept_ok(A, X, d) || !views || views->n_views < 1 || views->n_views > SSLREC_MAX_VIEWS) return SSLREC_E_BADARG;
    SweptArgs a = {};
    a.pack = A->pack; a.val = A->val; a.w_start = A->w_start; a.w_steps = A->w_steps;
    a.wf_ptr = A->wf_ptr; a.cf_ptr = A->cf_ptr; a.frow = A->f_row; a.fstart = A->f_start; a.fn = A->f_n;
    a.stamp = sslrec_take_stamp();
    a.n_slots = A->n_slots;
    a.X = X;
    a.n_views = views->n_views;
    a.eps = views->eps;
    if (views->scale_flags & ~(SSLREC_SCALE_Y | SSLREC_SCALE_ACC)) return SSLREC_E_BADARG;      // (the shared first product is a valued one)
    if (views->scale_flags && !views->row_scale) return SSLREC_E_BADARG;
    a.row_scale = views->scale_flags ? views->row_scale : nullptr;
    a.scale_flags = views->scale_flags;
    for (int k = 0; k < views->n_views; ++k) {
        if (!views->Y[k] && !views->acc_out[k]) return SSLREC_E_BADARG;
        if (views->acc_out[k] && !views->acc_in[k]) return SSLREC_E_BADARG;
        a.Y[k] = views->Y[k]; a.noise[k] = views->noise[k];
        a.acc_in[k] = views->acc_in[k]; a.acc_out[k] = views->acc_out[k];
        if (views->philox_noise[k]) {
            if (!views->philox || views->noise[k]) return SSLREC_E_BADARG;
            a.philox = views->philox;
            a.philox_stream[k] = views->philox_stream[k];
            a.philox_noise[k] = 1;
        }
    }
    return swept_dispatch(a, A, d, (hipStream_t)stream);
}
