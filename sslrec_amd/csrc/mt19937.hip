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
// One step of the generator is a LINEAR map of its 624 x 32 state bits over GF(2), so "advance by B blocks" is a bit
// matrix J = T^B (19,968 x 19,968 bits = 49.8 MB), whose column c is what B blocks make of the state with only bit c
// set -- the generator itself computes it, one workgroup per column (mt_basis_kernel, once per process).  With J, the
// states at blocks B, 2B, 3B, ... follow from the current one by one small matrix-vector product each (mt_apply_kernel:
// XOR of the columns whose bit is set), and then every stretch of B blocks has its own workgroup (mt_par_kernel).
__global__ __launch_bounds__(256) void mt_basis_kernel(long n_blocks, uint32_t *__restrict__ jump) {
    __shared__ uint32_t ring[MT_RING];
    const int c = blockIdx.x;                            // bit c of the state: word c / 32, bit c % 32
    for (int i = threadIdx.x; i < MT_N; i += 256) ring[i] = (i == c / 32) ? (1u << (c % 32)) : 0u;
    __shared__ uint32_t col[MT_N + 1];
    __syncthreads();
    mt_run<2>(ring, MT_N, col, nullptr, n_blocks * MT_N, 0.f);
    __syncthreads();
    for (int i = threadIdx.x; i < MT_N; i += 256) jump[(size_t)c * MT_N + i] = col[i];
}

// s_out ^= J s_in over this workgroup's 104 of the 19,968 columns (s_out zeroed beforehand; 192 workgroups)
#define MT_APPLY_GROUPS 192
#define MT_APPLY_COLS 104
__global__ __launch_bounds__(256) void mt_apply_kernel(const uint32_t *__restrict__ jump, const uint32_t *__restrict__ s_in,
                                                       uint32_t *__restrict__ s_out) {
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * MT_APPLY_COLS;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    for (int cw = c0 / 32; cw * 32 < c0 + MT_APPLY_COLS; ++cw) {
        uint32_t bits = s_in[cw];                        // uniform
        const int lo = c0 > cw * 32 ? c0 - cw * 32 : 0;
        const int hi = (c0 + MT_APPLY_COLS) < (cw + 1) * 32 ? c0 + MT_APPLY_COLS - cw * 32 : 32;
        bits &= (hi == 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        while (bits) {
            const int bbit = __ffs(bits) - 1;
            bits &= bits - 1;
            const uint32_t *colp = jump + (size_t)(cw * 32 + bbit) * MT_N;
            acc0 ^= colp[tid];
            acc1 ^= colp[tid + 256];
            if (tid + 512 < MT_N) acc2 ^= colp[tid + 512];
        }
    }
    if (acc0) atomicXor(&s_out[tid], acc0);
    if (acc1) atomicXor(&s_out[tid + 256], acc1);
    if (tid + 512 < MT_N && acc2) atomicXor(&s_out[tid + 512], acc2);
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
extern "C" size_t sslrec_mt19937_jump_bytes(void) { return (size_t)MT_N * 32 * MT_N * sizeof(uint32_t); }

extern "C" int sslrec_mt19937_jump_init(int64_t stretch_blocks, uint32_t *jump, void *stream) {
    if (stretch_blocks <= 0 || !jump) return SSLREC_E_BADARG;
    hipLaunchKernelGGL(mt_basis_kernel, dim3(MT_N * 32), dim3(256), 0, (hipStream_t)stream, (long)stretch_blocks, jump);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

// state_out[0..623] = the block `stretch_blocks` blocks after the block state_in[0..623] (one matrix-vector product over GF(2))
extern "C" int sslrec_mt19937_jump_apply(const uint32_t *jump, const uint32_t *state_in, uint32_t *state_out, void *stream) {
    if (!jump || !state_in || !state_out || state_in == state_out) return SSLREC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(state_out, 0, MT_N * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mt_apply_kernel, dim3(MT_APPLY_GROUPS), dim3(256), 0, st, jump, state_in, state_out);
    SSLREC_LAUNCH_CHECK();
    return 0;
}

static long mt_par_workers(int64_t stretch_blocks, int64_t n) {      // for an empty current block (the most workers)
    const long stretch = stretch_blocks * MT_N;
    return n <= stretch ? 1 : 1 + (n - stretch + stretch - 1) / stretch;
}

extern "C" size_t sslrec_mt19937_par_ws_bytes(int64_t stretch_blocks, int64_t n) {
    if (stretch_blocks <= 0 || n < 0) return 0;
    return (size_t)(mt_par_workers(stretch_blocks, n) + 2) * MT_N * sizeof(uint32_t) + 16;
}

static int mt_par_any(int mode, uint32_t *mt_state, const uint32_t *jump, int64_t stretch_blocks, uint32_t *ws, void *out, float keep_rate,
                      int64_t n, hipStream_t st) {
    if (!mt_state || !jump || !ws || stretch_blocks <= 0 || n <= 0 || !out) return SSLREC_E_BADARG;
    const long stretch = stretch_blocks * MT_N;
    const long workers = mt_par_workers(stretch_blocks, n);
    uint32_t *state_next = ws + (size_t)(workers + 1) * MT_N;          // 625 words
    if (workers > 1) {
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)workers * MT_N * sizeof(uint32_t), st);
        if (e != hipSuccess) return (int)e;
        for (long j = 1; j < workers; ++j)               // states[j] = J states[j - 1] (states[0] = the current block)
            hipLaunchKernelGGL(mt_apply_kernel, dim3(MT_APPLY_GROUPS), dim3(256), 0, st, jump, j == 1 ? mt_state : ws + (size_t)(j - 1) * MT_N,
                               ws + (size_t)j * MT_N);
        SSLREC_LAUNCH_CHECK();
    }
    if (mode == 0) hipLaunchKernelGGL(mt_par_kernel<0>, dim3((int)workers), dim3(256), 0, st, mt_state, ws, stretch, state_next, out, (long)n, keep_rate);
    else hipLaunchKernelGGL(mt_par_kernel<1>, dim3((int)workers), dim3(256), 0, st, mt_state, ws, stretch, state_next, out, (long)n, keep_rate);
    SSLREC_LAUNCH_CHECK();
    hipError_t e = hipMemcpyAsync(mt_state, state_next, (MT_N + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int sslrec_mt19937_uniform_par_f32(uint32_t *mt_state, const uint32_t *jump, int64_t stretch_blocks, uint32_t *ws, float *out,
                                              int64_t n, void *stream) {
    return mt_par_any(0, mt_state, jump, stretch_blocks, ws, out, 0.f, n, (hipStream_t)stream);
}

extern "C" int sslrec_mt19937_keep_mask_par(uint32_t *mt_state, const uint32_t *jump, int64_t stretch_blocks, uint32_t *ws, float keep_rate,
                                            uint8_t *keep_out, int64_t n, void *stream) {
    return mt_par_any(1, mt_state, jump, stretch_blocks, ws, keep_out, keep_rate, n, (hipStream_t)stream);
}
