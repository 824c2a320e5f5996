// The reference's augmentation draws, replayed on the device (parity mode without the host stall).
//
// EdgeDrop and EmbedPerturb draw with `t.rand(...)` on the global CPU generator and copy the numbers to the device
// (reference models/aug_utils.py:28,130); a step of SimGCL on the amazon-book-shaped graph needs 55 M of them, which the
// host generates in ~80 ms while the GPU step takes 3.6 ms.  The CPU generator is MT19937 and `t.rand` (float32) is
// (y & 0xFFFFFF) * 2^-24 of each tempered 32-bit output, in order (ATen: CPUGeneratorImpl::random() ->
// uniform_real_distribution<float>, a serial kernel).  The stream is sequential by construction, but one regeneration of
// the state is data-parallel 227 words at a time (a word needs the one 227 places back), so ONE workgroup advances the
// generator at about a word per clock: the same numbers, bit for bit, written straight into device memory.  The state (624 words + index of the next output) lives in device memory between calls; the host
// side (sslrec_amd/rng.py: HostGeneratorReplay) uploads it from `torch.get_rng_state()` and writes it back with
// `torch.set_rng_state()` whenever host code is about to draw.
#include "common.h"

#define MT_N 624
#define MT_M 397
// x[k] lives at ring[k & MT_RING_MASK].  Between two barriers a wave may be two rounds (454 words) ahead of another wave
// that still reads 624 words back: 1078 live words, so 1024 slots are NOT enough (a write of the second round would land on
// a word a slower wave has yet to read)
#define MT_RING 2048
#define MT_RING_MASK (MT_RING - 1)

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for the round's global stores (vmcnt),
// i.e. put a memory round trip on the critical path of every round
__device__ __forceinline__ void mt_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// MODE 0: out = float uniforms; MODE 1: out = uint8 keep mask, floor(u + keep_rate) != 0 (aug_utils.py:28-29)
template <int MODE>
__device__ __forceinline__ void mt_emit(void *out_, long o, uint32_t x, float keep_rate) {
    const float u = (float)(mt_temper(x) & 0xFFFFFFu) * 5.9604644775390625e-8f;      // * 2^-24, exact
    if constexpr (MODE == 0) reinterpret_cast<float *>(out_)[o] = u;
    else reinterpret_cast<uint8_t *>(out_)[o] = floorf(u + keep_rate) != 0.f ? 1 : 0;
}

// The generator as ONE sequence x[0], x[1], ...: x[k] = x[k-227] ^ twist(x[k-624], x[k-623]); the state in memory is the
// block x[0..623] and `pos` the next output inside it.  A round produces 227 consecutive words, thread t always the word
// k = 624 + 227 R + t: its x[k-227] is its OWN word of the round before (a register), and x[k-624], x[k-623] were written
// two or three rounds earlier, so two rounds run between barriers (the regeneration of a block in three dependent phases
// -- read, compute, write, barrier, read again -- took 1450 clocks per 624 words).  Output number o is x[pos + o].  At the end the state is the
// 624-word block holding the last consumed word, as the host generator would have it.
// MODE 2: nothing is written out (the generator is only advanced)
// ring: the workgroup's MT_RING words of LDS, holding x[0..623] on entry; returns with the block of the last consumed word
// written to state_out[0..623] and the next output index to state_out[624] (state_out may be null)
template <int MODE>
__device__ __forceinline__ void mt_run(uint32_t *ring, const long pos, uint32_t *__restrict__ state_out, void *__restrict__ out_,
                                       const long n, const float keep_rate) {
    const int tid = threadIdx.x;
    if constexpr (MODE != 2) {
        const long first = (MT_N - pos) < n ? (MT_N - pos) : n;
        for (long t = tid; t < first; t += 256) mt_emit<MODE>(out_, t, ring[pos + t], keep_rate);
    }
    if (pos + n <= MT_N) {                               // everything came out of the block at hand
        if (state_out) {
            for (int i = tid; i < MT_N; i += 256) state_out[i] = ring[i];
            if (tid == 0) state_out[MT_N] = (uint32_t)(pos + n);
        }
        return;
    }
    const long last = pos + n - 1;                       // sequence index of the last word consumed
    const long blk = last / MT_N;                        // its block (>= 1)
    const long rounds = (blk * MT_N + 226) / 227;        // words 624 .. 624 (blk + 1) - 1 must exist
    // Two rounds per barrier: x[k-624] and x[k-623] of round R were written in round R-3 or R-2, never in R-1, so the
    // rounds 2m and 2m+1 both read only what the barrier after round 2m-1 has already made visible.
    uint32_t prev = tid < 227 ? ring[397 + tid] : 0u;
    int at = (MT_N + tid) & MT_RING_MASK;                        // ring slot of this thread's word of the next round
    long o = MT_N + tid - pos;                           // ... and its output number
    for (long r = 0; r < rounds; r += 2) {
        const bool two = r + 1 < rounds;
        if (tid < 227) {
            const uint32_t a0 = ring[(at - 624) & MT_RING_MASK], b0 = ring[(at - 623) & MT_RING_MASK];
            const uint32_t a1 = ring[(at - 397) & MT_RING_MASK], b1 = ring[(at - 396) & MT_RING_MASK];
            const uint32_t w0 = prev ^ mt_twist(a0, b0);
            ring[at] = w0;
            if constexpr (MODE != 2) { if (o < n) mt_emit<MODE>(out_, o, w0, keep_rate); }
            prev = w0;
            if (two) {
                const uint32_t w1 = w0 ^ mt_twist(a1, b1);
                ring[(at + 227) & MT_RING_MASK] = w1;
                if constexpr (MODE != 2) { if (o + 227 < n) mt_emit<MODE>(out_, o + 227, w1, keep_rate); }
                prev = w1;
            }
            at = (at + 454) & MT_RING_MASK;
            o += 454;
        }
        mt_lds_barrier();
    }
    __syncthreads();
    if (state_out) {
        for (int i = tid; i < MT_N; i += 256) state_out[i] = ring[(blk * MT_N + i) & MT_RING_MASK];
        if (tid == 0) state_out[MT_N] = (uint32_t)(last + 1 - blk * MT_N);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void mt19937_kernel(uint32_t *__restrict__ state, void *__restrict__ out_, long n, float keep_rate) {
    __shared__ uint32_t ring[MT_RING];                      // x[k] lives at ring[k & MT_RING_MASK]: 624 words of look-back + 227 ahead fit
    for (int i = threadIdx.x; i < MT_N; i += 256) ring[i] = state[i];
    const long pos = (long)state[MT_N];                  // next output of the block; MT_N: block used up
    __syncthreads();
    mt_run<MODE>(ring, pos, state, out_, n, keep_rate);
}

// ---- many workgroups on one stream: jumping ahead ---------------------------------------------------------------------
// One step of the generator (the window x[k..k+623] moving on by one word) is a LINEAR map F of the window's bits over
// GF(2), and on windows the generator has produced phi(F) = 0 for its characteristic polynomial phi (degree 19937).  With
// g = x^J mod phi the window J words ahead is g(F) window = the XOR of the windows at the set bits of g -- windows of the
// next 19937 + 623 words only, whatever J is.  The host supplies the g's (sslrec_amd/mt_jump.py: two levels, worker
// fan1 k + j starts from one level-2 jump k and one level-1 jump j); mt_poly_apply_kernel generates those 20.5 k words in
// LDS and does the XOR: ~10 k set bits x 624 words of LDS reads per jump (81 us of one CU's LDS, spread over `nseg`
// workgroups that each take a slice of the polynomial).  Then every stretch of B blocks has its own workgroup
// (mt_par_kernel).  (Round 2 computed a 49.8 MB bit matrix T^B instead and chained one matrix-vector product per worker:
// 12 us each, one after the other -- 0.34 ms of a 9.2 M-number draw's 0.48 ms.)
#define MT_DEGREE 19937
#define MT_POLY_THREADS 1024

// dst ^= poly(F) src over this workgroup's slice of the polynomial's words (dst zeroed beforehand)
// level 2: application a = k - 1 (k = 1 ..): states[fan1 k] from the current block, polynomial n_l1 + a
// level 1: application a = k n_l1 + (j - 1): states[fan1 k + j] from states[fan1 k] (k = 0: the current block), polynomial j - 1
// level 0: one application: dst = ws from src = mt_state with polynomial 0 (sslrec_mt19937_jump_poly)
__global__ __launch_bounds__(MT_POLY_THREADS) void mt_poly_apply_kernel(const uint32_t *__restrict__ polys, const uint32_t *__restrict__ mt_state,
                                                                        uint32_t *__restrict__ ws, int fan1, int level, int workers, int nseg) {
    extern __shared__ uint32_t mt_x[];                   // x[0 ..]: the block, then the words after it (linear)
    __shared__ uint32_t red[MT_N];
    const int tid = threadIdx.x;
    const int app = blockIdx.x / nseg, seg = blockIdx.x % nseg;
    const int n_l1 = fan1 - 1;
    int src_idx, dst_idx, poly;
    if (level == 2) { src_idx = 0; dst_idx = fan1 * (app + 1); poly = n_l1 + app; }
    else if (level == 1) { const int k = app / n_l1, j = app % n_l1 + 1; src_idx = fan1 * k; dst_idx = src_idx + j; poly = j - 1; }
    else { src_idx = 0; dst_idx = 0; poly = 0; }
    if (level && dst_idx >= workers) return;
    const uint32_t *src = src_idx ? ws + (size_t)src_idx * MT_N : mt_state;
    uint32_t *dst = ws + (size_t)dst_idx * MT_N;
    const uint32_t *g = polys + (size_t)poly * MT_N;
    const int wps = (MT_N + nseg - 1) / nseg;            // polynomial words per slice
    const int w0 = seg * wps, w1 = min(w0 + wps, MT_N);
    if (w0 >= w1) return;
    const int i_end = min(w1 * 32, MT_DEGREE);           // windows 32 w0 .. i_end - 1: words up to x[i_end + 622]
    for (int i = tid; i < MT_N; i += MT_POLY_THREADS) { mt_x[i] = src[i]; red[i] = 0u; }
    __syncthreads();
    if (tid == 0) {
        // A seeded block is not a window the generator produced (31 low bits of x[0] are free; no later output depends on
        // them); put it on the subspace where phi(F) = 0 by giving x[0] the low bits x[623] ^ x[396] implies -- a no-op on
        // every generated block.
        uint32_t t = mt_x[623] ^ mt_x[396];
        const uint32_t lsb = t >> 31;
        t ^= lsb ? 0x9908b0dfu : 0u;
        mt_x[0] = (mt_x[0] & 0x80000000u) | ((t & 0x3fffffffu) << 1) | lsb;
    }
    __syncthreads();
    {                                                    // x[624 .. i_end + 623): 227 words per round, two rounds per barrier (see mt_run)
        const int need = i_end + 623;
        uint32_t prev = tid < 227 ? mt_x[397 + tid] : 0u;
        for (int k = MT_N + tid; k - tid < need; k += 454) {
            if (tid < 227) {
                const uint32_t w0_ = prev ^ mt_twist(mt_x[k - 624], mt_x[k - 623]);
                mt_x[k] = w0_;
                const uint32_t w1_ = w0_ ^ mt_twist(mt_x[k - 397], mt_x[k - 396]);
                mt_x[k + 227] = w1_;
                prev = w1_;
            }
            mt_lds_barrier();
        }
    }
    // thread = (quarter q of the slice's words, words j = r, r + 256, r + 512 of the window)
    const int q = __builtin_amdgcn_readfirstlane(tid >> 8), r = tid & 255;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    const bool third = r + 512 < MT_N;
    for (int w = w0 + q; w < w1; w += 4) {
        uint32_t bits = g[w];                            // wave-uniform
        if (w * 32 + 32 > MT_DEGREE) bits &= (1u << (MT_DEGREE - w * 32)) - 1u;      // (the polynomial's degree is < 19937 anyway)
        const uint32_t *xw = mt_x + w * 32 + r;
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            acc0 ^= xw[b];
            acc1 ^= xw[b + 256];
            if (third) acc2 ^= xw[b + 512];
        }
    }
    atomicXor(&red[r], acc0);
    atomicXor(&red[r + 256], acc1);
    if (third) atomicXor(&red[r + 512], acc2);
    __syncthreads();
    for (int i = tid; i < MT_N; i += MT_POLY_THREADS)
        if (red[i]) atomicXor(&dst[i], red[i]);
}

// worker j = blockIdx.x: from the state of block j*B (states[j], its own block already consumed unless j == 0) the next B
// blocks; worker 0 also emits what is left of the current block.  The grid is sized for an empty current block, so the
// last worker or two may find nothing to do (the launch is then the same whatever the position in the block: a captured
// hipGraph replays it).  The worker that emits output n - 1 writes the generator's new state to state_next.
template <int MODE>
__global__ __launch_bounds__(256) void mt_par_kernel(const uint32_t *__restrict__ state, const uint32_t *__restrict__ states, long stretch,
                                                     uint32_t *__restrict__ state_next, void *__restrict__ out_, long n, float keep_rate) {
    __shared__ uint32_t ring[MT_RING];
    const int j = blockIdx.x;
    const long pos0 = (long)state[MT_N];
    const long left0 = MT_N - pos0;                      // words still in the current block
    const long begin = j == 0 ? 0 : left0 + (long)j * stretch;          // first output of this worker
    if (begin >= n) return;
    const uint32_t *src = j == 0 ? state : states + (size_t)j * MT_N;
    for (int i = threadIdx.x; i < MT_N; i += 256) ring[i] = src[i];
    long count = (j == 0 ? left0 + stretch : stretch);
    if (begin + count > n) count = n - begin;
    const bool is_last = begin + count == n;
    __syncthreads();
    void *dst = MODE == 0 ? (void *)(reinterpret_cast<float *>(out_) + begin) : (void *)(reinterpret_cast<uint8_t *>(out_) + begin);
    mt_run<MODE>(ring, j == 0 ? pos0 : (long)MT_N, is_last ? state_next : nullptr, dst, count, keep_rate);
}

extern "C" int sslrec_mt19937_uniform_f32(uint32_t *mt_state, float *out, int64_t n, void *stream) {
    if (!mt_state || n < 0 || (n > 0 && !out)) return SSLREC_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mt19937_kernel<0>, dim3(1), dim3(256), 0, (hipStream_t)stream, mt_state, (void *)out, (long)n, 0.f);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sslrec_mt19937_keep_mask(uint32_t *mt_state, float keep_rate, uint8_t *keep_out, int64_t n, void *stream) {
    if (!mt_state || n < 0 || (n > 0 && !keep_out)) return SSLREC_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mt19937_kernel<1>, dim3(1), dim3(256), 0, (hipStream_t)stream, mt_state, (void *)keep_out, (long)n, keep_rate);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// ---- the same stream from many workgroups -----------------------------------------------------------------------------
static size_t mt_poly_lds_bytes() { return (size_t)(MT_DEGREE + MT_N + 2 * 454 + 64) * sizeof(uint32_t); }

static int mt_poly_launch(const uint32_t *polys, const uint32_t *mt_state, uint32_t *ws, int fan1, int level, int apps, int workers, hipStream_t st) {
    if (apps <= 0) return 0;
    const size_t lds = mt_poly_lds_bytes();
    static bool attr_set[64] = {};      // per device: the attribute belongs to the device's code object (and the call costs ~0.5 ms of host time:
    int dev = 0;                        // set on every launch it made a host-bound LightGCN parity step 1.57 instead of 0.59 ms)
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SSLREC_E_BADARG;
    if (!attr_set[dev]) {
        hipError_t ea = hipFuncSetAttribute((const void *)mt_poly_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return (int)ea;
        attr_set[dev] = true;
    }
    int nseg = 256 / apps;                               // about one workgroup per CU (256 / 512 / 1024 measured: no difference)
    nseg = nseg < 1 ? 1 : (nseg > 16 ? 16 : nseg);
    hipLaunchKernelGGL(mt_poly_apply_kernel, dim3(apps * nseg), dim3(MT_POLY_THREADS), lds, st, polys, mt_state, ws, fan1, level, workers, nseg);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// state_out[0..623] = poly(F) state_in[0..623]: the block J words on for poly = x^J mod phi (J a multiple of 624 to land on a block)
extern "C" int sslrec_mt19937_jump_poly(const uint32_t *poly, const uint32_t *state_in, uint32_t *state_out, void *stream) {
    if (!poly || !state_in || !state_out || state_in == state_out) return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(state_out, 0, MT_N * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    return mt_poly_launch(poly, state_in, state_out, 2, 0, 1, 1, st);
}

static long mt_par_workers(int64_t stretch_blocks, int64_t n) {      // for an empty current block (the most workers)
    const long stretch = stretch_blocks * MT_N;
    return n <= stretch ? 1 : 1 + (n - stretch + stretch - 1) / stretch;
}

extern "C" size_t sslrec_mt19937_par_ws_bytes(int32_t fan1, int32_t fan2) {
    if (fan1 < 2 || fan2 < 1) return 0;
    return (size_t)((size_t)fan1 * fan2 + 2) * MT_N * sizeof(uint32_t) + 16;
}

// polys: [(fan1 - 1) + (fan2 - 1)][624] words, x^(624 B j) for j = 1 .. fan1-1, then x^(624 B fan1 k) for k = 1 .. fan2-1
static int mt_par_any(int mode, uint32_t *mt_state, const uint32_t *polys, int fan1, int fan2, int64_t stretch_blocks, uint32_t *ws, void *out,
                      float keep_rate, int64_t n, hipStream_t st) {
    if (!mt_state || !polys || !ws || stretch_blocks <= 0 || fan1 < 2 || fan2 < 1 || n <= 0 || !out) return SSLREC_E_BADARG;
    const long stretch = stretch_blocks * MT_N;
    const long cap = (long)fan1 * fan2;                  // workers of one pass
    uint32_t *state_next = ws + (size_t)(cap + 1) * MT_N;            // 625 words
    for (int64_t done = 0; done < n;) {
        const int64_t n_c = (n - done) < cap * stretch ? (n - done) : cap * stretch;
        const long workers = mt_par_workers(stretch_blocks, n_c);   // <= cap
        if (workers > 1) {
            hipError_t e = hipMemsetAsync(ws, 0, (size_t)workers * MT_N * sizeof(uint32_t), st);
            if (e != hipSuccess) return (int)e;
            const int groups = (int)((workers + fan1 - 1) / fan1);  // level-2 targets: states[fan1 k], k = 1 .. groups - 1
            int rc = mt_poly_launch(polys, mt_state, ws, fan1, 2, groups - 1, (int)workers, st);
            if (rc) return rc;
            rc = mt_poly_launch(polys, mt_state, ws, fan1, 1, groups * (fan1 - 1), (int)workers, st);
            if (rc) return rc;
        }
        void *dst = mode == 0 ? (void *)((float *)out + done) : (void *)((uint8_t *)out + done);
        if (mode == 0) hipLaunchKernelGGL(mt_par_kernel<0>, dim3((int)workers), dim3(256), 0, st, mt_state, ws, stretch, state_next, dst, (long)n_c, keep_rate);
        else hipLaunchKernelGGL(mt_par_kernel<1>, dim3((int)workers), dim3(256), 0, st, mt_state, ws, stretch, state_next, dst, (long)n_c, keep_rate);
        SSLREC_LAUNCH_CHECK();
        hipError_t e = hipMemcpyAsync(mt_state, state_next, (MT_N + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
        done += n_c;
    }
    return 0;
}

extern "C" int sslrec_mt19937_uniform_par_f32(uint32_t *mt_state, const uint32_t *polys, int32_t fan1, int32_t fan2, int64_t stretch_blocks,
                                              uint32_t *ws, float *out, int64_t n, void *stream) {
    return mt_par_any(0, mt_state, polys, fan1, fan2, stretch_blocks, ws, out, 0.f, n, (hipStream_t)stream);
}

extern "C" int sslrec_mt19937_keep_mask_par(uint32_t *mt_state, const uint32_t *polys, int32_t fan1, int32_t fan2, int64_t stretch_blocks,
                                            uint32_t *ws, float keep_rate, uint8_t *keep_out, int64_t n, void *stream) {
    return mt_par_any(1, mt_state, polys, fan1, fan2, stretch_blocks, ws, keep_out, keep_rate, n, (hipStream_t)stream);
}
