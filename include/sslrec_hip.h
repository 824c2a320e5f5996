/*
 * sslrec_hip.h -- C ABI of libsslrec_hip.so, the MI355X (gfx950) implementation of
 * SSLRec's general-CF hot path.
 *
 * The reference (HKUDS/SSLRec) is pure Python and has NO FFI of its own: the path is a
 * sequence of PyTorch calls.  Each entry point below replaces one such call sequence;
 * the reference file:line it stands in for is cited on every declaration (paths are
 * relative to the reference root).  INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - nothing is allocated inside: the caller owns every buffer (workspace sizes come
 *     from the *_ws_bytes queries);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronizes;
 *   - return value is a hipError_t cast to int (0 = hipSuccess); argument errors
 *     return SSLREC_E_BADARG without launching anything;
 *   - all arithmetic is fp32; row indices into embedding tables are int64 (what
 *     trainer/trainer.py:64 puts on the device), graph indices are int32.
 */
#ifndef SSLREC_HIP_H
#define SSLREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSLREC_ABI_VERSION 7
#define SSLREC_E_BADARG 1001   /* distinct from any hipError_t */

int sslrec_abi_version(void);

/* bits[0 .. ceil(n_rows / 32)) = the set { idx0[i] + off0, idx1[i] + off1, idx2[i] + off2 : i < n } as a bitmap (idx1 / idx2 nullable);
 * n_rows <= 1,048,576 (one workgroup builds it in LDS).  The rows a fused BPR backward (sslrec_bpr_bwd_f32) writes of a stacked
 * [users; items] gradient table: anchors at offset 0, positives and negatives at offset n_user (lightgcn.py:49-52). */
int sslrec_row_bits3(const int64_t *idx0, int64_t off0, const int64_t *idx1, int64_t off1, const int64_t *idx2, int64_t off2,
                     int32_t n, int32_t n_rows, uint32_t *bits, void *stream);

/* ------------------------------------------------------------------------------------
 * Sparse propagation  Y = A * X      (replaces torch.spmm(adj, embeds),
 * models/general_cf/lightgcn.py:28-29; its autograd backward dX = A^T dY,
 * trainer/trainer.py:67; and LightGCL's gather/index_add_ product,
 * models/general_cf/lightgcl.py:58-65)
 *
 * A is held in a STREAMED, PACKED CSR built for ONE embedding size d.  Rows keep their
 * entries sorted by column (the summation order); rows longer than a cap are cut into chunks;
 * the row segments are dealt to `n_waves` work streams of (nearly) equal length -- 32 per
 * compute unit, one per resident wavefront.  A stream is a sequence of SLOTS: G = 256/d
 * consecutive slots form one vector LOAD (one 16-byte read per lane fetches G neighbour rows of
 * 4*d bytes each; slot k belongs to lane group k % G), every row segment occupies whole loads
 * (padded with col = -1, val = 0), 4 loads form a block and a block is stored group-major:
 *     element(k) = (k / 4G) * 4G + (k % G) * 4 + (k / G) % 4
 * so that a lane reads the 4 columns (values) it needs for a block with one 16-byte load.
 * Chunk partial sums go to a scratch slab and are combined in slot order by a second small
 * kernel (no atomics, bit-deterministic).
 * ---------------------------------------------------------------------------------- */
typedef struct sslrec_csr {
    int32_t n_rows, n_cols, nnz;   /* nnz = real entries (pads excluded)                   */
    int32_t d;                     /* embedding size the layout was packed for             */
    int32_t n_elem;                /* elements of col[] / val[] (pads included)            */
    const int32_t *col;        /* [n_elem] column of each slot, -1 = pad                  */
    const float   *val;        /* [n_elem]                                                */
    int32_t n_waves;           /* number of work streams                                  */
    const int32_t *w_start;    /* [n_waves]   first element of stream w (multiple of 4G)  */
    const int32_t *w_len;      /* [n_waves]   LOADS in stream w                           */
    const int32_t *r_ptr;      /* [n_waves+1] row segments of stream w = [r_ptr[w], r_ptr[w+1]) */
    int32_t n_rseg;
    const int32_t *r_len;      /* [n_rseg] LOADS of each row segment, in stream order     */
    const int32_t *r_dst;      /* [n_rseg] >=0: output row; <0: partial slot ~x           */
    int32_t n_long;            /* rows that were cut into chunks                          */
    const int32_t *long_row;   /* [n_long]                                                */
    const int32_t *long_ptr;   /* [n_long+1] slots of row i = [long_ptr[i],long_ptr[i+1]) */
    int32_t n_slots;           /* partial slab holds n_slots*d floats                     */
} sslrec_csr_t;                /* the struct itself lives in HOST memory                  */

/* ZERO-VALUED ENTRIES.  Pads carry the value 0 and an edge-dropped view of the streamed / row-bundled layouts may zero an
 * entry's value instead of removing it: there an element whose stored value is exactly 0 contributes NOTHING, even when the row it
 * points at holds Inf / NaN (a pad reads row 0).  The column-swept kernel removes dropped entries from its streams and marks pads by
 * their packed word (-1), so it has no such rule: an explicit zero-weight edge of the INPUT matrix is multiplied like any other
 * (0 * Inf = NaN, as torch.spmm gives).  The layouts therefore differ only on explicit zeros in the caller's value array whose
 * source row is non-finite -- the reference's adjacencies hold none (every value is 1 / sqrt(d_i d_j) > 0). */
/* Optional fused epilogue applied to each finished output row y (all pointers nullable):
 *   noise  : y += eps * sign(y) * noise_row / max(||noise_row||_2, 1e-12)
 *            (EmbedPerturb, models/aug_utils.py:125-132; noise is the caller's draw)
 *   acc_in/acc_out : acc_out_row = acc_in_row + y   (the layer SUM of lightgcn.py:41 in
 *            forward; the "+G" of the backward recurrence).  acc_in may equal acc_out. */
typedef struct sslrec_epilogue {
    const float *noise; float eps;
    const float *acc_in; float *acc_out;
    /* device-side noise (perf mode): with noise == NULL and philox != NULL the uniform noise row
     * is COMPUTED in the epilogue (Philox4x32-10, see "device-side augmentation RNG" below): no N x d tensor exists */
    const uint64_t *philox; uint32_t philox_stream;
    /* EmbedPerturb when the launch works on a COLUMN SLICE of the table (feature-sliced tables, sslrec_amd/feature_shard.py:
     * a GPU holds d/P columns of every row, but the perturbation's norm runs over the FULL row of the reference's [N, d] draw):
     *   noise_sumsq [n_rows] (nullable): squared L2 norm of the full noise row -- replaces the norm over the launch's own columns
     *                (supplied noise: the ranks all-reduce their partial sums of squares; Philox noise: every rank computes all d
     *                columns' draws of a row, sslrec_philox_row_sumsq);
     *   noise_row_stride / noise_col_off (floats, multiples of 4; 0 / 0 = the launch's own table): position of this launch's
     *                columns inside the full noise row -- the Philox element index of a computed draw is taken in the FULL table,
     *                so a slice of the result equals the same columns of the one-GPU result. */
    const float *noise_sumsq; int32_t noise_row_stride, noise_col_off;
    /* acc_out_row += axpy_alpha * (axpy_scale ? *axpy_scale : 1) * axpy_x_row (nullable; needs acc_out): the regularizer's gradient
     * 2 * reg_weight * g * E0 (loss_utils.py:20-24) folded into the LAST product of the backward recurrence instead of a pass of
     * its own over the table plus an elementwise add */
    const float *axpy_x; float axpy_alpha; const float *axpy_scale;
    /* HINT (nullable): x_row_bits[r / 32] bit (r % 32) clear = row r of X is all zeros.  The column-swept kernel then treats the
     * entries of that column like pads (no gather of the row, no accumulate); the other kernels ignore the hint.  The result is the
     * dense product's (x + 0 = x; a zero's sign aside).  Use: the first product of the backward recurrence when the incoming gradient
     * is the BPR loss's -- it touches <= 3B of the N rows (sslrec_row_bits3 builds the bitmap from the batch's indices). */
    const uint32_t *x_row_bits;
    /* DEFERRED layer sum (n_sum_in <= SSLREC_MAX_SUM_IN; needs acc_in and acc_out): acc_out_row = (((acc_in_row + sum_in[0]_row) + ...)
     * + sum_in[n-1]_row) + y.  The forward layer loop (lightgcn.py:38-41) then writes only Y in the launches l < L and the LAST launch
     * forms E0 + E1 + ... + E_L -- added in layer order, so the result has the bits of the running sum -- instead of every launch
     * reading and writing the running sum: 2 (L - 1) n_rows d 4 fewer bytes written per forward.  Column-swept kernel on layouts for
     * which sslrec_swept_deferred_sum_ok() returns 1 only (elsewhere n_sum_in != 0 is SSLREC_E_BADARG and the caller keeps the
     * running sum). */
    int32_t n_sum_in; const float *sum_in[3];
    /* FACTORIZED normalization (ABI 6; column-swept kernel only, elsewhere scale_flags != 0 is SSLREC_E_BADARG).  The reference's
     * adjacency is D^-1/2 A D^-1/2 with a 0/1 matrix A (data_handler_general_cf.py:37-51, binarized at :65), LightGCL's is
     * 1/sqrt(d_u d_i) (lightgcl.py:17-20): value(i, j) = r[i] * c[j].  So a layer chain (lightgcn.py:38-41) can carry the SCALED table
     * F_l = c (.) E_l as the gathered operand and never read the value stream:
     *   SSLREC_SCALE_PATTERN : every entry counts 1 -- the value array is not read; the row sum s of the gathered rows becomes the
     *                          product's row y = row_scale[row] * s before any other epilogue (noise, layer sum);
     *   SSLREC_SCALE_Y       : Y receives row_scale[row] * y (the next pattern launch's operand) -- acc_out still adds the plain y;
     *   SSLREC_SCALE_ACC     : acc_out receives row_scale[row] * (acc_in + y [+ axpy]) (the backward recurrence g <- G + A^T g carried
     *                          as c (.) g).
     * row_scale [n_rows] is needed by all three.  A chain = one valued launch with SCALE_Y (its operand E_0 is unscaled), then pattern
     * launches.  The products agree with the valued ones to rounding (r[i] * c[j] is rounded once more in the value array), not
     * bit for bit. */
    const float *row_scale; int32_t scale_flags;
} sslrec_epilogue_t;           /* host memory */
#define SSLREC_MAX_SUM_IN 3
#define SSLREC_SCALE_PATTERN 1
#define SSLREC_SCALE_Y 2
#define SSLREC_SCALE_ACC 4

/* d must be 32, 64, 128 or 256 and equal A->d.  Y may be NULL when only acc_out is wanted.
 * col/val/r_len/w_len default to A's arrays when the override pointers are NULL; the
 * overrides are how an edge-dropped or re-valued view (below) is multiplied (r_len and
 * w_len overrides come as a pair).
 * partial_ws: >= A->n_slots*d floats (may be NULL when n_slots==0). */
int sslrec_spmm_csr_f32(const sslrec_csr_t *A,
                        const int32_t *col_override, const float *val_override,
                        const int32_t *r_len_override, const int32_t *w_len_override,
                        const float *X, int32_t d, float *Y,
                        const sslrec_epilogue_t *epi, float *partial_ws, void *stream);

/* NARROW tables beyond the column-swept layout -- d = 8 or 16 (a GPU's d/P columns of every row under feature slicing,
 * sslrec_amd/feature_shard.py; BASELINE config 5: 10 M-row tables, 128 / 8 = 16 columns) and optionally 32: the ROW-BUNDLED
 * streamed layout.  Replaces the same reference call (lightgcn.py:28-29 / lightgcl.py:58-65).  At these widths one
 * 16-byte-per-lane wave instruction gathers G = 256/d = 32 / 16 / 8 neighbour rows, so every lane group (d/4 lanes) owns a
 * whole OUTPUT row and keeps its sum in registers: no cross-lane reduction, no LDS, control stays wave-uniform.  Row
 * segments (a row, or a chunk of a row longer than the cap: partial sums to the slab, combined in slot order by the
 * long-row kernel) are sorted by length and G consecutive ones form a BUNDLE that runs for the longest member's steps
 * (rounded up to S = d/4 steps; shorter members are padded with col = -1: 2-5 % pads on power-law graphs of mean degree 32);
 * bundles are dealt (longest-processing-time-first) to n_waves streams.  A stream is stored like a column-swept stream:
 * 64-dword blocks of S steps, dword g*(64/G) + j of a block = the entry of (step j, lane group g), so ONE coalesced dword
 * load per array fetches S steps for all G rows and a DPP quad / row permute hands every lane its entry.
 *   col/val [n_elem]; w_ptr [n_waves+1] bundles of stream w (stored back to back from element w_start[w]);
 *   b_steps [n_bundles] steps of a bundle (multiple of S, may be 0: rows without entries are written as zeros);
 *   b_dst [n_bundles * G]: output row of lane group g (>= 0), partial slot ~x (< 0), or SSLREC_BUNDLE_NONE. */
#define SSLREC_BUNDLE_NONE (-2147483647 - 1)
typedef struct sslrec_bundled {
    int32_t n_rows, n_cols, nnz, d;
    int32_t n_elem;
    const int32_t *col; const float *val;
    int32_t n_waves;
    const int32_t *w_start;    /* [n_waves]   first element of stream w (multiple of 64)            */
    const int32_t *w_ptr;      /* [n_waves+1]                                                       */
    int32_t n_bundles;
    const int32_t *b_steps;    /* [n_bundles]                                                       */
    const int32_t *b_dst;      /* [n_bundles * 256/d]                                               */
    int32_t n_long;
    const int32_t *long_row, *long_ptr;
    int32_t n_slots;
} sslrec_bundled_t;            /* host memory; arrays on the device */

/* val_override / b_steps_override (nullable): a re-valued view / a view whose bundles were shortened.  d must equal A->d.
 * partial_ws: >= A->n_slots * d floats (NULL when n_slots == 0). */
int sslrec_spmm_bundled_f32(const sslrec_bundled_t *A, const float *val_override, const float *X, int32_t d, float *Y,
                            const sslrec_epilogue_t *epi, float *partial_ws, void *stream);
/* the same product over a VIEW of the layout: col / val / b_steps / w_blocks overrides (all nullable; b_steps and w_blocks together, with
 * col: a compacted edge-dropped view, sslrec_bundled_compact) */
int sslrec_spmm_bundled_view_f32(const sslrec_bundled_t *A, const int32_t *col_override, const float *val_override,
                                 const int32_t *b_steps_override, const int32_t *w_blocks_override, const float *X, int32_t d, float *Y,
                                 const sslrec_epilogue_t *epi, float *partial_ws, void *stream);

/* Column-swept variant of the same product for output tables that fit the chip's LDS
 * (n_rows * d * 4 <= n_blocks * SSLREC_SWEPT_LDS_BYTES; amazon-book at d=64: 36.9 MB of 40 MB).
 * Replaces the same reference call (lightgcn.py:28-29) with the same epilogue.  One workgroup per CU
 * owns `n_slots` LDS accumulators (a slot = an output row, or one interleaved chunk of a heavy row);
 * its 16 waves x (256/d) lane groups own disjoint slots and walk their edges sorted by column, so the
 * CUs of an XCD sweep X together (sslrec_amd/csrc/spmm_swept.hip).
 *   pack/val [n_elem]: per wave a stream of 64-dword blocks of S = min(16, 64/G) steps, G = 256/d
 *                      lane groups: dword g*(64/G) + j of a block = the entry of (step j, lane group g), repeated in every
 *                      16-lane row of the lane group at d >= 128 (one coalesced dword load per array fetches S steps, a DPP
 *                      broadcast hands every lane its entry); pack = column (low 20 bits) | slot << 20, -1 = pad;
 *                      w_start [16*n_blocks] element offsets, w_steps steps (multiples of S)
 *   flush records (f_row global row, f_start first slot, f_n slots to add): a row that occupies ONE slot belongs to one
 *                      lane group, hence to one wave, which writes it out as soon as its own sweep is over (no barrier; the
 *                      stores overlap the slower waves' gathers): records [wf_ptr[w], wf_ptr[w+1]) of wave w = 16*block + wave;
 *                      rows cut into chunks are added up by the whole block after its barrier: records [cf_ptr[b], cf_ptr[b+1])
 * d (= A->d) is 32, 64, 128 or 256, or -- for FEATURE-SLICED tables, where a GPU holds d/P columns of every row
 * (sslrec_amd/feature_shard.py) -- 16 or 8 (beyond this layout's size limits those run on the row-bundled layout above). */
#define SSLREC_SWEPT_LDS_BYTES 163840
typedef struct sslrec_swept {
    int32_t n_rows, n_cols, nnz, d;
    int32_t n_elem, n_blocks, n_slots;
    const int32_t *pack; const float *val;
    const int32_t *w_start, *w_steps;
    const int32_t *wf_ptr;                 /* [16*n_blocks+1] */
    const int32_t *cf_ptr;                 /* [n_blocks+1]    */
    const int32_t *f_row, *f_start, *f_n;  /* [n_rows]        */
} sslrec_swept_t;              /* host memory; arrays on the device */

/* pack_override / val_override [n_elem] and w_steps_override [16*n_blocks] (all nullable) multiply an edge-dropped
 * or re-valued view. */
/* d = the tables' embedding size: A->d, or 2 / 4 / 8 times A->d -- then the product runs as d / A->d launches, one per
 * block of A->d embedding columns of the [N, d] tables (a table wider than the LDS holds: sslrec_plan_layout falls back
 * to such a layout before it falls back to the streamed kernel). */
int sslrec_swept_deferred_sum_ok(const sslrec_swept_t *A);      /* 1: launches on A take sslrec_epilogue_t.sum_in (see there) */
int sslrec_spmm_swept_f32(const sslrec_swept_t *A, const int32_t *pack_override, const float *val_override,
                          const int32_t *w_steps_override, const float *X, int32_t d, float *Y,
                          const sslrec_epilogue_t *epi, void *stream);

/* One product, several epilogues: SimGCL's three views (simgcl.py:29-31: two perturbed forwards + a clean one)
 * start from the SAME A.E0, so their first layer is one launch whose flush applies up to SSLREC_MAX_VIEWS
 * (noise, acc_in, acc_out, Y) sets; noise[k] NULL = clean view.  (The mirror image holds for the gradient of that
 * layer: A^T applied once to the sum of the views' gradients.) */
#define SSLREC_MAX_VIEWS 4
typedef struct sslrec_epilogue_views {
    int32_t n_views; float eps;
    float *Y[SSLREC_MAX_VIEWS];
    const float *noise[SSLREC_MAX_VIEWS];
    const float *acc_in[SSLREC_MAX_VIEWS];
    float *acc_out[SSLREC_MAX_VIEWS];
    const uint64_t *philox;                      /* device-side noise for the views with philox_noise[k] != 0 */
    uint32_t philox_stream[SSLREC_MAX_VIEWS];
    int32_t philox_noise[SSLREC_MAX_VIEWS];
    const float *row_scale; int32_t scale_flags;   /* SSLREC_SCALE_Y / SSLREC_SCALE_ACC for every view (see sslrec_epilogue_t; ABI 6) */
} sslrec_epilogue_views_t;     /* host memory */
int sslrec_spmm_swept_views_f32(const sslrec_swept_t *A, const float *X, int32_t d,
                                const sslrec_epilogue_views_t *views, void *stream);

/* EdgeDrop on the swept layout (replaces EdgeDrop.forward, models/aug_utils.py:18-31): keep[] is the
 * reference's per-entry mask in the ORIGINAL COO order, edge_map[e] the COO entry that governs element e
 * (unused for pads).  Every lane group's stream is compacted inside its own slots: kept entries (values times
 * scale) move to the front in their original order, the rest becomes pads; w_steps_out receives the new step
 * counts (longest compacted stream of each wave, rounded up to 4). */
int sslrec_swept_compact(const sslrec_swept_t *A, const int32_t *edge_map, const uint8_t *keep, float scale,
                         int32_t *pack_out, float *val_out, int32_t *w_steps_out, void *stream);

/* Device-side augmentation RNG (perf mode; the reference draws on the CPU generator, models/aug_utils.py:28,130, and
 * the parity mode of this library consumes exactly those draws).  philox_state = 2 x uint64 in DEVICE memory:
 * {seed, step}.  A uniform is a pure function of (seed, step, stream, element): `stream` is a per-call constant chosen
 * by the host (distinct per augmentation call of a training step), element = the COO entry id (EdgeDrop: forward and
 * transposed views of one call therefore agree) or the row-major float index / 4 of the output table (EmbedPerturb);
 * u = (x >> 8) * 2^-24.  sslrec_philox_advance does step += 1 on the device -- call it once per training step; it is a
 * kernel, so a captured hipGraph draws fresh numbers on every replay. */
int sslrec_philox_advance(uint64_t *philox_state, void *stream);
/* the noise of call `philox_stream` written out: out[4i .. 4i+3] = the four uniforms of float group i (n a multiple of 4).
 * What the epilogues compute on the fly; for tests and for callers that want the dense EmbedPerturb tensor. */
int sslrec_philox_fill_f32(const uint64_t *philox_state, uint32_t philox_stream, float *out, size_t n, void *stream);
/* out[r] = sum over the d draws of row r of that noise squared (d a multiple of 4): the full-row norm of EmbedPerturb for a
 * launch that only holds a column slice of the row */
int sslrec_philox_row_sumsq(const uint64_t *philox_state, uint32_t philox_stream, int32_t n_rows, int32_t d, float *out, void *stream);
/* out[r] (+)= sum_c x[r, c]^2 for a row-major [n_rows, d] table (d a multiple of 4): a rank's share of the same norm for SUPPLIED noise */
int sslrec_row_sumsq_f32(const float *x, int32_t n_rows, int32_t d, float *out, void *stream);
/* The reference's own draws on the device (parity mode): the CPU generator behind `t.rand` (models/aug_utils.py:28,130) is
 * MT19937; mt_state = uint32[625] in DEVICE memory = its 624 state words + the index of the next output in the current
 * block (624 = block exhausted), as found in torch.get_rng_state().  The calls write the next n numbers of that stream --
 * (y & 0xFFFFFF) * 2^-24 per tempered output, bit-identical to `t.rand(n)` on the host -- and leave mt_state advanced by
 * n.  One workgroup (the stream is sequential); capturable in a hipGraph. */
int sslrec_mt19937_uniform_f32(uint32_t *mt_state, float *out, int64_t n, void *stream);
/* the same stream turned into EdgeDrop's mask: keep_out[i] = floor(u_i + keep_rate) != 0 */
int sslrec_mt19937_keep_mask(uint32_t *mt_state, float keep_rate, uint8_t *keep_out, int64_t n, void *stream);
/* The same stream from many workgroups.  One step of MT19937 is a linear map F of the state bits over GF(2); with
 * g = x^J mod phi (phi = the generator's characteristic polynomial, degree 19937) the state J words ahead is g(F) state =
 * the XOR of the 624-word windows at the set bits of g over the NEXT 19937 + 623 words.  The host computes the g's
 * (sslrec_amd/mt_jump.py; a polynomial = uint32[624], bit i = coefficient of x^i); the *_par calls take a two-level
 * table -- polys[(fan1-1) + (fan2-1)][624]: x^(624 B j) for j = 1..fan1-1, then x^(624 B fan1 k) for k = 1..fan2-1 --
 * derive the states at blocks B, 2B, ... (two launches) and give every stretch of B blocks its own workgroup: the same
 * numbers as the one-workgroup calls, the same state afterwards, any n (passes of fan1*fan2 stretches).
 * ws: sslrec_mt19937_par_ws_bytes(fan1, fan2) bytes. */
int sslrec_mt19937_jump_poly(const uint32_t *poly, const uint32_t *state_in, uint32_t *state_out, void *stream);
size_t sslrec_mt19937_par_ws_bytes(int32_t fan1, int32_t fan2);
int sslrec_mt19937_uniform_par_f32(uint32_t *mt_state, const uint32_t *polys, int32_t fan1, int32_t fan2, int64_t stretch_blocks,
                                   uint32_t *ws, float *out, int64_t n, void *stream);
int sslrec_mt19937_keep_mask_par(uint32_t *mt_state, const uint32_t *polys, int32_t fan1, int32_t fan2, int64_t stretch_blocks,
                                 uint32_t *ws, float keep_rate, uint8_t *keep_out, int64_t n, void *stream);
/* EdgeDrop with the mask computed in place: entry k is kept iff floor(u_k + keep_rate) != 0 (aug_utils.py:28-29) */
int sslrec_swept_compact_philox(const sslrec_swept_t *A, const int32_t *edge_map, float keep_rate,
                                const uint64_t *philox_state, uint32_t philox_stream, float scale,
                                int32_t *pack_out, float *val_out, int32_t *w_steps_out, void *stream);
/* EdgeDrop on the row-bundled layout: val_out[e] = keep(edge_map[e]) ? A->val[e] * scale : 0 -- the mask given (keep != NULL, one byte per
 * entry) or computed (keep == NULL: Philox, as above).  Hand val_out to sslrec_spmm_bundled_f32 as `val_override`: a zero-valued element
 * contributes exactly nothing (no compaction: the view runs the full stream length).  edge_map: sslrec_plan_edge_map(plan, d, STREAMED). */
int sslrec_bundled_drop_values(const sslrec_bundled_t *A, const int32_t *edge_map, const uint8_t *keep, float keep_rate,
                               const uint64_t *philox_state, uint32_t philox_stream, float scale, float *val_out, void *stream);
/* The same views COMPACTED (round 5; replaces EdgeDrop.forward, models/aug_utils.py:18-31, on the row-bundled layout): every row of a
 * bundle keeps its kept entries in their order, a bundle runs for the longest of its compacted rows and the bundles of a stream move up
 * behind each other, so a keep-0.5 view gathers about half of what the zero-valued form gathers.  col_out / val_out [A->n_elem],
 * b_steps_out [A->n_bundles], w_blocks_out [A->n_waves] (64-element blocks of every stream still in use) go to
 * sslrec_spmm_bundled_view_f32 as the four overrides.  Sums equal the zero-valued form's bit for bit (same entries, same order). */
int sslrec_bundled_compact(const sslrec_bundled_t *A, const int32_t *edge_map, const uint8_t *keep, float scale, int32_t *col_out,
                           float *val_out, int32_t *b_steps_out, int32_t *w_blocks_out, void *stream);
int sslrec_bundled_compact_philox(const sslrec_bundled_t *A, const int32_t *edge_map, float keep_rate, const uint64_t *philox_state,
                                  uint32_t philox_stream, float scale, int32_t *col_out, float *val_out, int32_t *b_steps_out,
                                  int32_t *w_blocks_out, void *stream);
int sslrec_edge_drop_compact_philox(const sslrec_csr_t *A, const int32_t *edge_map, float keep_rate,
                                    const uint64_t *philox_state, uint32_t philox_stream, float scale,
                                    int32_t *col_out, float *val_out, int32_t *r_len_out, int32_t *w_len_out, void *stream);

/* Edge dropout without rebuilding the matrix (replaces EdgeDrop.forward,
 * models/aug_utils.py:18-31: boolean-index values/indices, rebuild COO).
 * keep[k] (uint8, 0/1) is the reference's per-entry mask in the ORIGINAL COO entry order;
 * edge_map[e] gives, for element e of col[], the COO entry whose mask bit governs it (the
 * entry itself for the forward matrix, the transposed entry for the backward matrix; unused
 * for pads).  Kept entries of every row segment are re-packed into consecutive slots of the
 * same stream (same w_start, same element mapping) in col_out / val_out [n_elem];
 * r_len_out / w_len_out receive the new lengths in loads (a fully dropped row keeps a
 * zero-length segment and is written as zeros);
 * scale multiplies kept values (1/keep_rate when EdgeDrop(resize_val=True), else 1). */
int sslrec_edge_drop_compact(const sslrec_csr_t *A, const int32_t *edge_map,
                             const uint8_t *keep, float scale,
                             int32_t *col_out, float *val_out, int32_t *r_len_out,
                             int32_t *w_len_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Native plan builder (host C++ inside the library; sslrec_amd/csrc/plan.cpp).  The reference converts its
 * uncoalesced COO adjacency (data_utils/data_handler_general_cf.py:70-73) to CSR inside EVERY torch.spmm call
 * (models/general_cf/lightgcn.py:28-29); here a caller hands over the matrix once -- as COO entries in any order or
 * as plain (rowptr, col, val) -- and receives the sslrec_swept_t / sslrec_csr_t layouts above.
 *   build_coo : entries sorted by (row, column), stable (duplicates kept in input order = the summation order)
 *   build_csr : entries of a row keep the order they are given in
 *   layout    : builds, on the host, the layout for embedding size d.  kind AUTO picks the column-swept layout when
 *               the output table fits the chip's LDS and the streamed one otherwise; returns the kind built (> 0)
 *               or a negative error.  flags: SSLREC_PLAN_NO_XCD_SPLIT keeps the two row classes of a bipartite
 *               adjacency on all XCDs.
 *   host_array: named host arrays of a layout ("pack","val","w_start","w_steps","wf_ptr","cf_ptr","f_row","f_start","f_n",
 *               "edge_map","elem_host","csr_pos_host" (f_ptr of ABI 2 became "wf_ptr" + "cf_ptr") / "col","val","w_start","w_len","r_ptr","r_len","r_dst","long_row",
 *               "long_ptr","edge_map",...; d = 0: "rowptr","col","val","perm" of the CSR) for callers that manage
 *               device memory themselves (the Python host does, through PyTorch's allocator)
 *   upload    : hipMalloc + copy of a layout; afterwards swept()/csr()/edge_map() return device-side descriptors and
 *               sslrec_plan_spmm_f32 multiplies with whichever kernel the layout belongs to.
 * edge_map[e] = index of the INPUT entry behind element e (-1 for pads): feed it to sslrec_swept_compact /
 * sslrec_edge_drop_compact with a per-input-entry keep mask. */
typedef struct sslrec_plan sslrec_plan_t;      /* opaque, host memory */
#define SSLREC_PLAN_AUTO 0
#define SSLREC_PLAN_SWEPT 1
#define SSLREC_PLAN_STREAMED 2
#define SSLREC_PLAN_BUNDLED 3      /* what kind STREAMED builds at d = 8 / 16 (and at 32 on request): reported by sslrec_plan_info */
#define SSLREC_PLAN_NO_XCD_SPLIT 1
typedef struct sslrec_plan_info {
    int32_t n_rows, n_cols; int64_t nnz;
    int32_t kind, d, xcd_split;
    int32_t n_elem, n_blocks, n_slots, n_streams, n_rseg, n_long;
    int64_t xcd_col_pairs;      /* swept layout with the XCD split: sum over the 8 XCDs of the DISTINCT columns their output rows reference
                                   (x 4 d bytes = what the XCDs' L2s pull through the fabric per launch when every line is fetched once per XCD) */
} sslrec_plan_info_t;
int sslrec_plan_build_coo(const int64_t *rows, const int64_t *cols, const float *vals, int64_t nnz, int32_t n_rows,
                          int32_t n_cols, sslrec_plan_t **out);                                  /* host pointers */
int sslrec_plan_build_csr(const int64_t *rowptr, const int32_t *col, const float *val, int32_t n_rows, int32_t n_cols,
                          sslrec_plan_t **out);                                                  /* host pointers */
int sslrec_plan_set_option(sslrec_plan_t *p, const char *name, int64_t value);   /* "seg_max": chunk cap of long rows in the
                                                        streamed layout; "n_streams": its number of work streams (0 = automatic);
                                                        "swept_blocks" 0 / 256 / 512; "xcd_balance" per mille; "swept_passes" 0 / 1:
                                                        allow a swept layout of d/2, d/4 ... columns run in embedding-column passes;
                                                        "xcd_cluster": row -> XCD co-clustering of the swept layout (rows that share
                                                        columns on the same XCD; same results bit for bit, only the fabric traffic
                                                        changes): 0 = never, 1..16 = always, that many refinement passes, 17 = automatic
                                                        (the default): kept when it lowers the distinct (XCD, column) pairs by > 25 % */
int sslrec_plan_layout(sslrec_plan_t *p, int32_t d, int32_t kind, int32_t flags);
/* the calls below address the layout of (d, kind); kind AUTO = the swept layout when one was built, else the streamed */
int sslrec_plan_info(const sslrec_plan_t *p, int32_t d, int32_t kind, sslrec_plan_info_t *info);
int sslrec_plan_host_array(const sslrec_plan_t *p, int32_t d, int32_t kind, const char *name, const void **ptr, int64_t *count,
                           int32_t *elem_bytes);
int sslrec_plan_upload(sslrec_plan_t *p, int32_t d, int32_t kind, void *stream);
const sslrec_swept_t *sslrec_plan_swept(const sslrec_plan_t *p, int32_t d);
const sslrec_csr_t *sslrec_plan_csr(const sslrec_plan_t *p, int32_t d);
const sslrec_bundled_t *sslrec_plan_bundled(const sslrec_plan_t *p, int32_t d);
const int32_t *sslrec_plan_edge_map(const sslrec_plan_t *p, int32_t d, int32_t kind);
int sslrec_plan_spmm_f32(const sslrec_plan_t *p, int32_t d, const float *X, float *Y, const sslrec_epilogue_t *epi,
                         void *stream);
void sslrec_plan_free(sslrec_plan_t *p);

/* Diagnostic, not an operator (no reference counterpart): while enabled, every launch of the column-swept kernel records per
 * wave the 100 MHz wall clock at the start of each metadata block (slots 0..28), at the end of its sweep (29) and at the
 * start (30) and end (31) of its flush, for the last 4 launches.  enable != 0 with host_out == NULL starts recording for
 * layouts of n_waves = 16 * n_blocks waves; a call with host_out copies the ring [4][n_waves][32] out; enable == 0 stops and
 * frees.  Returns the number of launches recorded so far (tools/spmm_trace.py; EXPERIMENTS.md B 4.1b's time line). */
int sslrec_debug_swept_trace(int enable, unsigned long long *host_out, int n_waves);

/* Measurement hook, not an operator: in-kernel launch timing that survives hipGraph capture (HIP events cannot be recorded
 * inside a captured graph's kernels, and a replayed step is what bench.py --gpus N times).  `record` = 4 x uint64 in DEVICE
 * memory, initialised to {~0, 0, 0, 0}; the NEXT SpMM launch of this thread (either kernel; all column passes of one call)
 * is handed the record: every workgroup stamps the 100 MHz wall clock when it starts (minimum kept in record[0]) and when
 * it has finished (its stores drained); the last one to finish adds (its clock - the minimum) to record[1] and 1 to record[3],
 * and resets record[0].  So after K replays record[1] / record[3] is the kernel's average duration in 10 ns ticks, first
 * workgroup start to last workgroup end -- the quantity rocprofv3's kernel trace reports.  NULL cancels a pending record. */
int sslrec_debug_stamp_next_launch(unsigned long long *record);
/* rate of that wall clock on the current device in kHz (hipDeviceAttributeWallClockRate; 100000 on MI355X), <= 0 on error */
int sslrec_debug_wall_clock_khz(void);

/* ------------------------------------------------------------------------------------
 * BPR loss over gathered rows (replaces the three gathers + cal_bpr_loss,
 * models/general_cf/lightgcn.py:49-52 and models/loss_utils.py:7-10; variant 1 is
 * LightGCL's -log(sigmoid(pos-neg)), models/general_cf/lightgcl.py:106-108).
 * Ta/Tp/Tn are row-major [*, d] tables; ia/ip/in are int64 row ids or NULL (= row b).
 *   fwd: loss_out[0] = sum_b f(<a_b,n_b> - <a_b,p_b>) / divisor  (divisor = the batch size folds the caller's `/ B`,
 *        lightgcn.py:52; 1 for the plain sum);   ws: sslrec_bpr_ws_bytes(B) bytes that MUST BE ZERO at entry -- it starts with ticket
 *        counters: the workgroup that finishes last adds the partial sums (fixed order) inside the same launch; the call leaves the
 *        counters zero again, so a workspace zeroed once can be reused call after call (not by two streams at the same time)
 *   fwd_total: the same launch also writes loss_out2[1] = loss_out2[0] + add_in[0] (add_in: a device scalar, e.g. the
 *        regularizer term: `bpr_loss + reg_loss` of lightgcn.py:54 without an elementwise launch of its own)
 *   bwd: dTa[ia[b]] += g*..., etc.  With an index array the contributions are added DETERMINISTICALLY (per destination
 *        row in ascending sample order: bit-reproducible, duplicates allowed, tables may alias) using
 *        ws = sslrec_bpr_bwd_ws_bytes(B, d) bytes; ws == NULL or 3B > 16384 falls back to atomic adds.  Without an
 *        index array: a plain store to row b.  gscale = upstream gradient (already divided by B when the caller averages). */
size_t sslrec_bpr_ws_bytes(int32_t B);
size_t sslrec_bpr_bwd_ws_bytes(int32_t B, int32_t d);
int sslrec_bpr_fwd_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                       const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant,
                       float divisor, float *ws, float *loss_out, void *stream);
int sslrec_bpr_fwd_total_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                             const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant,
                             float divisor, const float *add_in, float *ws, float *loss_out2, void *stream);
int sslrec_bpr_bwd_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                       const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant, float divisor,
                       const float *gscale_dev, float *dTa, float *dTp, float *dTn, void *ws, void *stream);
/* The same backward on a workspace that is KEPT between calls (same B and d, one stream at a time): sslrec_bpr_bwd_table_init clears
 * the scatter table inside ws once; sslrec_bpr_bwd_kept_f32 then needs no clearing launch -- the reduction hands every slot it used
 * back cleared -- two launches per call instead of three.  3B <= 16384, d <= 256, ws != NULL.
 * zero_table / zero_elems (nullable / 0; all three roles indexed, 16-byte aligned, a multiple of 4 floats): the gradient table(s) the
 * rows are added into -- one contiguous range covering dTa, dTp, dTn -- zeroed by the staging launch itself instead of by a fill
 * launch of the caller's (37 MB at amazon-book size: it runs under the staging's latency). */
int sslrec_bpr_bwd_table_init(void *ws, int32_t B, int32_t d, void *stream);
int sslrec_bpr_bwd_kept_f32(const float *Ta, const int64_t *ia, const float *Tp, const int64_t *ip,
                            const float *Tn, const int64_t *in, int32_t B, int32_t d, int32_t variant, float divisor,
                            const float *gscale_dev, float *dTa, float *dTp, float *dTn, void *ws, float *zero_table,
                            size_t zero_elems, void *stream);

/* ------------------------------------------------------------------------------------
 * InfoNCE against ALL rows of a view (replaces cal_infonce_loss,
 * models/loss_utils.py:30-39, called at simgcl.py:49 / sgl.py:57-59; variant 1 is
 * LightGCL's un-normalized form, models/general_cf/lightgcl.py:114-118).
 *
 *   variant 0:  x^ = x / sqrt(1e-8 + |x|^2) for e1, e2, all
 *               loss = sum_b [ -<e1^_b,e2^_b>/temp + log sum_j exp(<e1^_b, all^_j>/temp) ]
 *   variant 1:  no normalization;
 *               loss = sum_b [ -clamp(<e1_b,e2_b>/temp,-5,5) + log(sum_j exp(<e1_b,all_j>/temp) + 1e-8) ]
 *               (the caller divides by B for the reference's .mean()).
 * e1 = T1[i1[b]], e2 = T2[i2[b]] (index arrays nullable = row b); all = ALL[0..M).
 * The B x M score matrix is never materialized: FP32 MFMA tiles, exp and row sums stay
 * in registers.  d must be 32, 64 or 128.
 *   fwd: loss_out[0] = loss;  ws (sslrec_infonce_ws_bytes) keeps the normalized operands and
 *        the B row sums; the SAME ws must be handed to bwd.
 *   bwd: dE1,dE2 are dense [B,d] (caller scatters), dALL is dense [M,d]; all three are
 *        overwritten.  gscale_dev: upstream gradient scalar on device.
 * `variant` carries three fields, and the forward and backward calls of one evaluation must pass the SAME value:
 *   bits 0..7    0 / 1 as above;
 *   bits 8..15   arithmetic of the B x M products (SSLREC_INFONCE_PREC_*; 0 = the process default: the environment variable
 *                SSLREC_INFONCE_PRECISION, else h3 for variant 0 and x6 for variant 1; h3 -- by default or by name -- runs as x6 when
 *                temp < log2(e) / 15.5 = 0.0931, where its exponent bias would go negative).  x6 = operands as three bf16 planes, six bf16-MFMA terms per product with fp32
 *                accumulation (fp32-level error, held to the fp32 tolerances by every parity test); fp32 = v_mfma_f32_32x32x2_f32;
 *                the other modes drop terms and are opt-in;
 *   bit 16       SSLREC_INFONCE_FWD_W: the forward call also accumulates the anchor-gradient sums W_b = sum_j exp(s_bj) all_j
 *                (they depend neither on the upstream gradient nor on the row sums) in the workspace, from the SAME score tiles
 *                its row sums come from, and the backward call does not recompute them: the B x M score products are formed
 *                twice per forward + backward instead of three times.  Set it when a backward call will follow (a caller that
 *                only wants the loss leaves it clear: the forward then costs half as much). */
#define SSLREC_INFONCE_PREC_DEFAULT 0
#define SSLREC_INFONCE_PREC_X6 1
#define SSLREC_INFONCE_PREC_FP32 2
#define SSLREC_INFONCE_PREC_X36 3
#define SSLREC_INFONCE_PREC_X3 4
#define SSLREC_INFONCE_PREC_X63 5
#define SSLREC_INFONCE_PREC_X6A 6
#define SSLREC_INFONCE_PREC_H3 7        /* two fp16 planes / three fp16-MFMA terms per product (22-bit operands; variant 0 only, variant 1 runs x6) */
#define SSLREC_INFONCE_FWD_W (1 << 16)
size_t sslrec_infonce_ws_bytes(int32_t B, int32_t M, int32_t d);
int sslrec_infonce_fwd_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                           int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                           int32_t variant, float *ws, float *loss_out, void *stream);
int sslrec_infonce_bwd_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                           int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                           int32_t variant, float *ws, const float *gscale_dev,
                           float *dE1, float *dE2, float *dALL, void *stream);

/* Backward with the scatter of the gathered rows' gradients inside (the index_put backward of `T1[i1]` / `T2[i2]`, simgcl.py:32-37):
 * dE [2B, d] receives dE1 then dE2; for a role with an index array the rows are ADDED into dT1 / dT2 (the caller's gradient
 * tables: zero-initialised, or -- T2 being `all` itself, the call shape of simgcl.py:49 / sgl.py:57-59 -- dT2 = dALL), duplicates
 * in ascending sample order (bit-reproducible).  The rows are registered by the launch that writes dE, one reduction launch adds
 * both roles; the table
 * (scatter_ws: sslrec_scatter_ws_bytes(2B) bytes) is cleared by the backward's first kernel.  2B <= 16384. */
int sslrec_infonce_bwd_scatter_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                                   int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                                   int32_t variant, float *ws, const float *gscale_dev, float *dE,
                                   float *dT1, float *dT2, float *dALL, void *scatter_ws, void *stream);

/* The same InfoNCE with `all` ROW-SHARDED over ranks (the reference has no multi-GPU
 * path -- this is the sharded form of loss_utils.py:30-39).  Every rank passes the same B anchor/positive
 * rows and ITS M rows of `all`; Z_b = sum_j exp(.) and W_b = sum_j exp(.) a_j are sums over j, so:
 * (`variant` as above, the same value in all four calls; with SSLREC_INFONCE_FWD_W step 1 also leaves this rank's W partials in ws
 * and step 3 only forms dALL and sums them.)
 *   1. shard_rowsum:      z_part[B]   = partial row sums over the local rows      -> host all-reduces z
 *   2. shard_loss:        loss_out[0] = sum_b(-pos_b + log z_total[b])  (same value on every rank)
 *   3. shard_bwd:         dALL[M,d] (complete for the local rows), w_part[B,d]     -> host all-reduces w
 *   4. shard_finish_bwd:  dE1,dE2 [B,d] from w_total (same values on every rank)
 * ws: sslrec_infonce_ws_bytes(B, M, d) with the LOCAL M, carried through the four calls. */
int sslrec_infonce_shard_rowsum_f32(const float *T1, const int64_t *i1, const float *T2, const int64_t *i2,
                                    int32_t B, const float *ALL, int32_t M, int32_t d, float temp,
                                    int32_t variant, float *ws, float *z_part, void *stream);
int sslrec_infonce_shard_loss_f32(int32_t B, int32_t M, int32_t d, int32_t variant, float *ws,
                                  const float *z_total, float *loss_out, void *stream);
int sslrec_infonce_shard_bwd_f32(int32_t B, int32_t M, int32_t d, float temp, int32_t variant, float *ws,
                                 const float *gscale_dev, float *w_part, float *dALL, void *stream);
int sslrec_infonce_shard_finish_bwd_f32(int32_t B, int32_t M, int32_t d, float temp, int32_t variant,
                                        float *ws, const float *gscale_dev, const float *w_total,
                                        float *dE1, float *dE2, void *stream);

/* Sum of squares of a parameter table and its gradient (replaces `W.norm(2).square()` per parameter in
 * reg_params, models/loss_utils.py:20-24): out[0] = weight * sum_i x_i^2 ;  dx = 2 * gscale * weight * x  (weight folds
 * the caller's `reg_weight *`, lightgcn.py:53; 1 for the plain sum).
 * x and dx must be 16-byte aligned; ws: sslrec_sumsq_ws_bytes() bytes, zero at entry (once: the call leaves its ticket counters zero
 * again -- the one-launch reduction of sslrec_bpr_fwd_f32). */
size_t sslrec_sumsq_ws_bytes(void);
int sslrec_sumsq_fwd_f32(const float *x, size_t n, float weight, float *ws, float *out, void *stream);
int sslrec_sumsq_bwd_f32(const float *x, size_t n, float weight, const float *gscale_dev, float *dx, void *stream);

/* Step-level helpers of the one-node contrastive steps (ops.contrastive_step: SimGCL simgcl.py:39-55, SGL sgl.py:45-65 as ONE autograd
 * node with a hand-written backward): the scalar arithmetic and table additions PyTorch would run as stock elementwise launches.
 *   weighted_sum4 : out6[0] = ((w0 t0 + w1 t1) + w2 t2) + w3 t3, out6[1 + k] = w_k t_k, out6[5] = w1 t1 + w2 t2 (the loss and its logged
 *                   parts, simgcl.py:50-54: bpr, the two contrastive terms, the regularizer; t_k: device scalars, nullable = 0)
 *   scalar_scale2 : out2 = (a x, b x) of a device scalar x (the upstream gradient times the terms' weights)
 *   add_tables    : out = (a + b) [+ c] elementwise (c nullable; n a multiple of 4, 16-byte aligned; out may alias an input) */
int sslrec_weighted_sum4_f32(const float *t0, float w0, const float *t1, float w1, const float *t2, float w2, const float *t3, float w3,
                             float *out6, void *stream);
int sslrec_scalar_scale2_f32(const float *x, float a, float b, float *out2, void *stream);
int sslrec_add_tables_f32(const float *a, const float *b, const float *c, float *out, size_t n, void *stream);

/* Adam step over one parameter tensor (replaces torch.optim.Adam as the reference's Trainer uses it,
 * trainer/trainer.py:45-49,68).  state: 4 floats on the device, zero-initialised:
 * sslrec_adam_tick advances the step count t kept in state[0] and stores lr/(1-beta1^t), sqrt(1-beta2^t)
 * (computed in double); sslrec_adam_apply_f32 then does, per element and in one pass,
 *   g' = g + weight_decay*p;  m += (1-beta1)(g'-m);  v = beta2 v + (1-beta2) g'^2;
 *   p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps)
 * (hyper-parameters as doubles: 1-beta is rounded to fp32 once, from the double, like PyTorch does).
 * Call tick once per optimizer step, apply once per tensor (16-byte aligned). */
int sslrec_adam_tick(float *state, double lr, double beta1, double beta2, void *stream);
int sslrec_adam_apply_f32(float *p, const float *g, float *m, float *v, size_t n, const float *state,
                          double beta1, double beta2, double eps, double weight_decay, void *stream);

/* Rank-q products of LightGCL's SVD view (replaces `u_mul_s @ (vt @ E)` and its autograd backward,
 * models/general_cf/lightgcl.py:83-84; q <= 16).  M(k, n) = M[k*stride_q + n*stride_n] addresses a row-major [q,N]
 * factor (stride_q = N, stride_n = 1) or a row-major [N,q] factor (stride_q = 1, stride_n = q).
 *   reduce: out[q,d] = sum_n M(.,n) X[n,:]   (two-stage, fixed order: deterministic; ws: sslrec_rankq_ws_bytes)
 *   expand: Y[n,:]   = sum_k M(k,n) S[k,:] */
size_t sslrec_rankq_ws_bytes(int32_t q, int32_t d);
int sslrec_rankq_reduce_f32(const float *M, int64_t stride_q, int64_t stride_n, const float *X, int32_t N, int32_t d,
                            int32_t q, float *ws, float *out, void *stream);
int sslrec_rankq_expand_f32(const float *M, int64_t stride_q, int64_t stride_n, const float *S, int32_t N, int32_t d,
                            int32_t q, float *Y, void *stream);

/* ------------------------------------------------------------------------------------
 * All-rank evaluation: top-k unseen items per user (replaces full_predict + _mask_predict + t.topk,
 * models/general_cf/lightgcn.py:58-66, models/base_model.py:35-36, trainer/metrics.py:99-103).
 *   scores[u, i] = <UE[users[u]], IE[i]> in exact-fp32 MFMA tiles (d = 32, 64 or 128; from 2048 users on, on two fp16 planes per table with
 *   22-bit operands -- SSLREC_EVAL_PRECISION=fp32 / h3 forces either); items of the user's train row
 *   (trn_rowptr [n_user_rows + 1], trn_col sorted inside a row, int64, device; both NULL = nothing seen) are skipped --
 *   the reference gives them -1e8, i.e. they lose to every unseen item;  out_idx [n_users, k] int64, descending by
 *   score, ties by ascending item id, -1 where a user has fewer than k unseen items; out_val (nullable) the scores.
 *   users (nullable = rows 0..n_users-1): int64 row ids into UE.  k <= 64.  ws: sslrec_eval_topk_ws_bytes bytes.
 * Neither the [B, I] score matrix nor the dense train mask the reference copies from the host ever exists. */
/* full_predict + _mask_predict themselves (models/general_cf/lightgcn.py:58-66, models/base_model.py:35-36) -- the dense matrix is the
 * reference's plugin contract: out[u, i] = s (1 - m) - 1e8 m with s = <UE[users[u]], IE[i]> (exact-fp32 MFMA tiles, d = 32 / 64 / 128)
 * and m = train_mask[u, i] ([n_users, n_items] row-major, nullable = no mask; mask_elem_bytes 8 = int64 as the reference's .long()
 * batch, 4 = float32, 1 = uint8 / bool).  One pass: the mask is read once, the result written once, no intermediate matrix. */
int sslrec_full_predict_f32(const float *UE, const int64_t *users, int32_t n_users, const float *IE, int32_t n_items, int32_t d,
                            const void *train_mask, int32_t mask_elem_bytes, float *out, void *stream);
size_t sslrec_eval_topk_ws_bytes(int32_t n_users, int32_t n_items, int32_t k);
int sslrec_eval_topk_f32(const float *UE, const int64_t *users, int32_t n_users, const float *IE, int32_t n_items,
                         int32_t d, const int64_t *trn_rowptr, const int64_t *trn_col, int32_t k, void *ws,
                         int64_t *out_idx, float *out_val, void *stream);

/* Measurement aid (csrc/micro.hip; bench.py's `roofline.ceiling`): `blocks` workgroups of 16 waves gather random rows of the
 * [n_rows, row_bytes] table X (row_bytes 128 / 256 / 512; 8 loads in flight per wave, `iters` rounds); *n_gathers = rows fetched.
 * scratch: >= 16 KiB, never written. */
int sslrec_debug_gather_rows(const float *X, uint32_t n_rows, int32_t row_bytes, int32_t iters, int32_t blocks, float *scratch,
                             int64_t *n_gathers, void *stream);

/* Negative sampling (replaces PairwiseTrnData.sample_negs, data_utils/datasets_general_cf.py:13-20; SURVEY.md §8f
 * rank 3): negs_out[i] = an item drawn uniformly from [0, n_item) and redrawn while (users[i], item) is a train
 * interaction (trn_rowptr / trn_col as above).  Draws come from the Philox stream (philox_state, philox_stream) --
 * the reference's distribution, not its numpy random stream. */
int sslrec_sample_negs(const int64_t *users, int64_t n, const int64_t *trn_rowptr, const int64_t *trn_col, int32_t n_item,
                       const uint64_t *philox_state, uint32_t philox_stream, int64_t *negs_out, void *stream);

/* The SAME negative sampling on the host, continuing numpy's global MT19937 generator bit for bit (ABI 7): the
 * reference's loop `np.random.randint(item_num)` until `(u, iNeg) not in dokmat`, one interaction after the other
 * (data_utils/datasets_general_cf.py:13-20), consumes one 32-bit generator output per attempt (masked rejection).
 * HOST pointers throughout; no device work.  mt_key[624] / *mt_pos = `np.random.get_state()[1:3]`, advanced in place
 * (write them back with `np.random.set_state`); users int32 [n] (coomat.row); trn_rowptr int64 [n_user+1] / trn_col int32
 * ascending within a row = the train interactions (the dok matrix's keys); negs_out int32 [n]; n_draws (nullable) = generator
 * outputs consumed.  SSLREC_E_BADARG when a user has interacted with every item (the reference would not return). */
int sslrec_sample_negs_mt19937(uint32_t *mt_key, int32_t *mt_pos, const int32_t *users, int64_t n,
                               const int64_t *trn_rowptr, const int32_t *trn_col, int32_t n_user, int32_t n_item,
                               int32_t *negs_out, int64_t *n_draws);

/* rows of src [B,d] are added into dst[idx[b], :] (the index_put backward of the gathers at lightgcn.py:49-51 /
 * simgcl.py:32-37), duplicates in a fixed order (per destination in ascending b: bit-reproducible) with
 * ws = sslrec_scatter_ws_bytes(B) bytes; ws == NULL or B > 16384: atomic adds. */
size_t sslrec_scatter_ws_bytes(int32_t B);
int sslrec_scatter_add_rows_f32(const float *src, const int64_t *idx, int32_t B, int32_t d,
                                float *dst, void *ws, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSLREC_HIP_H */
