"""torch.autograd bindings of the HIP kernels (libsslrec_hip.so through ctypes).

PyTorch is used for device memory, streams and autograd bookkeeping only; every
arithmetic step of the hot path runs in the hand-written gfx950 kernels.  There is NO
CPU or eager-PyTorch fallback: tensors must live on a HIP device and the shared object
must be built, otherwise these functions raise.
"""
import ctypes as C

import os

import torch

from . import _lib
from .graph import BundledLayout, DroppedView, PropGraph, RevaluedView, graph_of

# When set to a list, every SpMM launch appends (start_event, end_event, plan, d, has_acc): the
# measurement hook bench.py uses to time the dominant kernel with HIP events on the launch stream.
PROFILE = None
# An event record is a packet of its own on the launch stream (a few microseconds of bubble each): PROFILE_EVERY = n records a pair
# around every n-th SpMM launch only (n coprime with the launches of a step, so the sampled launch rotates through all of them); the
# other launches are still listed, with None for the two events
PROFILE_EVERY = 1
_profile_tick = 0


def _profile_this_launch():
    """False: the hook is off, or this launch is listed without events (PROFILE_EVERY); True: an event pair goes around it"""
    global _profile_tick
    if PROFILE is None:
        return False
    _profile_tick += 1
    return PROFILE_EVERY <= 1 or _profile_tick % PROFILE_EVERY == 0


# The same hook for the fused InfoNCE: when set to a list, every forward / backward call of _InfoNceFn appends
# (start_event, end_event, 'fwd' | 'bwd', B, M, d, variant word incl. precision and SSLREC_INFONCE_FWD_W) -- bench.py's MFMA roofline
# of the steps InfoNCE dominates (SimGCL, SGL)
PROFILE_INFONCE = None

# In-kernel launch timing for steps that are REPLAYED from a captured hipGraph (HIP events cannot be recorded inside one):
# when set to a StampLog, every SpMM launch is handed a 4 x uint64 device record in which the kernel itself accumulates its
# duration by the device's wall clock (include/sslrec_hip.h: sslrec_debug_stamp_next_launch); the record's address is baked
# into the captured launch, so after K replays it holds the sum over K executions.
STAMPS = None


class StampLog:
    def __init__(self, device, capacity=512):
        self.buf = torch.zeros((capacity, 4), dtype=torch.int64, device=device)
        self.buf[:, 0] = -1                      # running minimum of the start clocks
        self.meta = []
        self.khz = int(_lib.load().sslrec_debug_wall_clock_khz())

    def attach_next(self, *meta):
        if len(self.meta) >= self.buf.shape[0]:
            raise RuntimeError('StampLog is full (%d launches)' % self.buf.shape[0])
        rc = _lib.load().sslrec_debug_stamp_next_launch(self.buf[len(self.meta)].data_ptr())
        _lib.check(rc, 'sslrec_debug_stamp_next_launch')
        self.meta.append(meta)

    def reset_counts(self):
        """forget the executions so far (warm-up replays): sums and counts back to zero"""
        torch.cuda.synchronize(self.buf.device)
        self.buf[:, 1:] = 0
        self.buf[:, 0] = -1
        torch.cuda.synchronize(self.buf.device)

    def read(self):
        """[(meta, average duration in ms, executions)] of every launch that ran at least once"""
        host = self.buf.cpu().numpy()
        out = []
        for i, meta in enumerate(self.meta):
            n = int(host[i, 3])
            if n > 0:
                out.append((meta, float(host[i, 1]) / n / self.khz, n))
        return out

SPMM_DIMS = (32, 64, 128, 256)
# sslrec_epilogue_t.scale_flags (include/sslrec_hip.h): the factorized normalization of a layer chain on the column-swept kernel
SCALE_PATTERN, SCALE_Y, SCALE_ACC = 1, 2, 4
# SSLREC_SPMM_FACTORIZED=0: every launch of a layer chain reads the value stream (the round-1..4 form; bit-compatible with them).
# Default: when the adjacency's values factorize as r[i] * c[j] (PropGraph.factorization: the reference's D^-1/2 A D^-1/2 does), the
# chain carries the scaled table and only its first launch per direction reads values -- equal to rounding, not bit for bit.
# Chains with EmbedPerturb (SimGCL) keep the valued form unless SSLREC_SPMM_FACTORIZED=2: sign(y) (aug_utils.py:130) is discontinuous at 0,
# the valued chain adds the reference's own fp32 products in the reference's order (bit-equal to torch.spmm on almost every row), the
# factorized one rounds every term differently -- at amazon-book size that flipped the sign of about one element in 5.5e7 (call b:
# 201 gradient elements off by 1e-9 in the whole-step test), i.e. one perturbed element 0.2 away from the reference's.
FACTORIZED = {'0': 0, '1': 1, '2': 2}.get(os.environ.get('SSLREC_SPMM_FACTORIZED', '1'), 1)


def _chain_scale(adj, d, layer_num, perturbed=False):
    """row factor r [N] when the layer chain  E_l = A E_{l-1} / g_{l-1} = G + A^T g_l  over `adj` can run factorized: a plain or
    edge-dropped (values kept) view of a PropGraph whose values are r[i] * r[j], column-swept layouts in both directions, L >= 2;
    `perturbed`: the chain carries the EmbedPerturb epilogue (factorized only at level 2)"""
    if not FACTORIZED or layer_num < 2 or (perturbed and FACTORIZED < 2):
        return None
    if isinstance(adj, DroppedView):
        if adj.scale != 1.0:
            return None
        graph = adj.graph
    elif isinstance(adj, PropGraph):
        graph = adj
    else:
        return None
    if graph.bwd is None:
        return None
    fact = graph.factorization()
    if fact is None or not fact[2]:
        return None
    if graph.fwd.swept(d) is None or graph.bwd.swept(d) is None:
        return None
    return fact[0]
MAX_SUM_IN = 3          # SSLREC_MAX_SUM_IN: earlier layers' tables the last forward launch can add up (deferred layer sum)
# SSLREC_DEFERRED_SUM=1 (opt-in): the forward layer loop writes only E_l in the launches l < L and sums E0..E_L in the last one
# (swept layouts, 2 <= L <= 4; bit-identical to the running sum).  Measured in round 4 (EXPERIMENTS.md): 74 MB fewer bytes written per
# forward at amazon-book size, but the step is no faster (0.545-0.551 against 0.540-0.541 ms eager, 0.494-0.496 against 0.496-0.501 ms
# as one hipGraph) -- the last launch's flush waits for three table rows per output row instead of one -- so the running sum stays
# the default.
DEFERRED_SUM = os.environ.get('SSLREC_DEFERRED_SUM', '0') == '1'
# narrow tables (a GPU's d / P columns under feature slicing, sslrec_amd/feature_shard.py): column-swept kernel, and the
# row-bundled streamed kernel beyond that layout's size limits
SPMM_NARROW_DIMS = (8, 16)
INFONCE_DIMS = (32, 64, 128)
EVAL_KMAX = 64          # largest k of the fused evaluation kernel (csrc/eval.hip: per-user key buffers in LDS)
# arithmetic of the InfoNCE products, carried in bits 8..15 of the C ABI's `variant` (include/sslrec_hip.h);
# None = the process default (SSLREC_INFONCE_PRECISION, else h3 for the normalized variant 0 and x6 for LightGCL's variant 1)
INFONCE_PRECISIONS = {None: 0, 'x6': 1, 'fp32': 2, 'x36': 3, 'x3': 4, 'x63': 5, 'x6a': 6, 'h3': 7}
# SSLREC_INFONCE_FWD_W (bit 16 of `variant`): a forward that autograd will differentiate also accumulates the anchor-gradient sums
# W = sum_j exp(s_bj) all_j from the score tiles its row sums come from, and the backward does not recompute them (the B x M score
# products of a forward + backward: two instead of three).  A forward under no_grad / on tensors without requires_grad runs the
# plain row-sum kernel.  SSLREC_INFONCE_FWD_W=0 restores the three-pass form (A/B measurements).
INFONCE_FWD_W_BIT = 1 << 16
INFONCE_FWD_W = os.environ.get('SSLREC_INFONCE_FWD_W', '1') != '0'


def _variant_code(variant, precision):
    if precision not in INFONCE_PRECISIONS:
        raise ValueError('unknown InfoNCE precision %r (one of %s)' % (precision, sorted(k for k in INFONCE_PRECISIONS if k)))
    return int(variant) | (INFONCE_PRECISIONS[precision] << 8)


def _stream():
    return torch.cuda.current_stream().cuda_stream


_TICKET_WS = {}


def _ticket_ws(device, n_bytes, kind):
    """workspace of the one-launch reductions (sslrec_bpr_fwd_f32, sslrec_sumsq_fwd_f32): its first word is a ticket counter that must be 0
    at entry and that every call leaves 0 -- zeroed ONCE here and reused by every call of that kind on the device (a fresh torch.empty
    per call would need a fill launch per call; a hipGraph capture reuses the tensor its eager warm-up steps made, so no fill is
    captured either).  Calls on one stream are ordered; when the calling stream changes, the new stream first waits for the old one."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), kind)
    cur = torch.cuda.current_stream(dev)
    ent = _TICKET_WS.get(key)
    if ent is None or ent[0].numel() * 4 < n_bytes:
        ent = _TICKET_WS[key] = [torch.zeros(max(n_bytes // 4, 1), dtype=torch.float32, device=dev), cur]
    elif ent[1] != cur:
        if not torch.cuda.is_current_stream_capturing():      # (a capture starts behind a device-wide synchronize)
            cur.wait_stream(ent[1])
        ent[1] = cur
    return ent[0]


import collections
import threading

_KEPT_WS = collections.OrderedDict()      # (device, B, d) -> [workspace, stream]; least recently used first
_KEPT_WS_MAX = 8                          # (~1.7 MB + 12 B d bytes each -- variable batch sizes must not pile them up)
_KEPT_WS_LOCK = threading.Lock()          # (two Python threads must not initialise / evict at once; a workspace still serves one stream at a time)
SSLREC_KEPT_SCATTER = os.environ.get('SSLREC_KEPT_SCATTER', '1') != '0'      # 0: a fresh workspace + a clearing launch per BPR backward


def _bpr_bwd_ws(device, B, d):
    """(workspace, kept): the BPR backward's staging rows + scatter table.  Kept per (device, B, d) and initialised once
    (sslrec_bpr_bwd_table_init): every call hands the table back clean, so a call is two launches instead of three.  Like the ticket
    workspaces it serves one stream at a time (a change of stream waits for the old one); batches beyond the table's 16,384
    contributions, or d > 256, take a fresh workspace and the ABI's own fallback."""
    lib = _lib.load()
    n = lib.sslrec_bpr_bwd_ws_bytes(B, d) // 4 + 1
    if not SSLREC_KEPT_SCATTER or 3 * B > 16384 or d > 256 or B <= 0:
        return torch.empty(n, dtype=torch.float32, device=device), False
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(B), int(d))
    cur = torch.cuda.current_stream(dev)
    with _KEPT_WS_LOCK:
        ent = _KEPT_WS.get(key)
        if ent is None:
            ws = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(lib.sslrec_bpr_bwd_table_init(ws.data_ptr(), int(B), int(d), _stream()), 'sslrec_bpr_bwd_table_init')
            ent = _KEPT_WS[key] = [ws, cur]
            while len(_KEPT_WS) > _KEPT_WS_MAX and not torch.cuda.is_current_stream_capturing():
                #  the dropped workspace was allocated on one stream and may last have been used on another (ent[1]): tell the
                # caching allocator, or it could hand the block back to the allocating stream while a backward on the other still reads it
                _k, old_ent = _KEPT_WS.popitem(last=False)
                if old_ent[1] is not None:
                    old_ent[0].record_stream(old_ent[1])
        else:
            _KEPT_WS.move_to_end(key)
            if ent[1] != cur:
                if not torch.cuda.is_current_stream_capturing():
                    cur.wait_stream(ent[1])
                ent[1] = cur
    return ent[0], True


def _check_kept(rc, what, device, B, d):
    """like _lib.check; a kept-workspace call that failed may have left its scatter table dirty: forget the workspace, the next call
    initialises a fresh one"""
    if rc != 0:
        dev = torch.device(device)
        with _KEPT_WS_LOCK:
            _KEPT_WS.pop((dev.index if dev.index is not None else torch.cuda.current_device(), int(B), int(d)), None)
    _lib.check(rc, what)


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('sslrec_amd kernels run on a HIP device only (got a %s tensor); there is no CPU '
                               'fallback by design' % t.device)


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError('fp32 tensors expected, got %s' % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def _idx(t):
    if t is None:
        return None
    if t.dtype != torch.int64:
        t = t.long()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ----------------------------------------------------------------------------------------------
# raw launcher
# ----------------------------------------------------------------------------------------------
def spmm_raw(adj, x, which='fwd', y=None, noise=None, eps=0.0, acc_in=None, acc_out=None, want_y=True, chained=False,
             noise_sumsq=None, noise_geom=None, axpy=None, x_row_bits=None, sum_in=None, row_scale=None, scale_flags=0):
    """Launch one CSR SpMM with optional fused epilogue.  `adj` is a PropGraph or DroppedView;
    `which` selects A ('fwd') or A^T ('bwd').  Returns y (or None when want_y=False).  (`chained` is accepted and ignored.)
    Column slices of a table (feature-sliced tables): `noise_sumsq` [n_rows] = squared norm of the FULL noise row, `noise_geom`
    = (columns of the full table, first column of this slice) for the element index of computed (Philox) draws.
    `axpy` = (x [n_rows, d], alpha, scale tensor or None): acc_out += alpha * scale * x, fused (the regularizer's gradient).
    `x_row_bits` = RowBits or None: a hint that the rows of x outside the bitmap are all zeros (sslrec_epilogue_t.x_row_bits).
    `sum_in` = up to 3 tables: acc_out = ((acc_in + sum_in[0]) + ...) + y (deferred layer sum; column-swept layouts only).
    `row_scale` [n_rows] + `scale_flags` (SCALE_PATTERN | SCALE_Y | SCALE_ACC): the factorized normalization of a layer chain
    (sslrec_epilogue_t.scale_flags; column-swept layouts only)."""
    view = adj if isinstance(adj, (DroppedView, RevaluedView)) else None
    graph = adj.graph if view is not None else adj
    plan = getattr(graph, which)
    _need_gpu(x)
    x = _f32c(x)
    n, d = x.shape
    if n != plan.n_cols:
        raise ValueError('operand has %d rows, matrix has %d columns' % (n, plan.n_cols))
    if d not in SPMM_DIMS and d not in SPMM_NARROW_DIMS:
        raise ValueError('embedding size %d not supported by the raw HIP SpMM launcher (supported: %s and %s); the ops.spmm / '
                         'ops.propagate_sum wrappers zero-pad other sizes' % (d, SPMM_NARROW_DIMS, SPMM_DIMS))
    if want_y and y is None:
        y = torch.empty((plan.n_rows, d), dtype=torch.float32, device=x.device)
    epi = None
    keep_alive = []
    if noise is not None or acc_out is not None or x_row_bits is not None or scale_flags:
        epi = _lib.EpilogueStruct()
        if scale_flags:
            if row_scale is None or row_scale.numel() != plan.n_rows or row_scale.dtype != torch.float32 or not row_scale.is_contiguous():
                raise ValueError('row_scale: contiguous fp32 [%d] expected' % plan.n_rows)
            epi.row_scale, epi.scale_flags = row_scale.data_ptr(), int(scale_flags)
        if x_row_bits is not None:
            if x_row_bits.n_rows != n:
                raise ValueError('row bitmap of %d rows for an operand of %d rows' % (x_row_bits.n_rows, n))
            epi.x_row_bits = x_row_bits.bits.data_ptr()
        if noise_sumsq is not None:
            keep_alive.append(_f32c(noise_sumsq))
            epi.noise_sumsq = keep_alive[-1].data_ptr()
        if noise_geom is not None:
            epi.noise_row_stride, epi.noise_col_off = int(noise_geom[0]), int(noise_geom[1])
        if axpy is not None:
            ax, alpha, scale = axpy
            keep_alive.append(_f32c(ax))
            if tuple(keep_alive[-1].shape) != (plan.n_rows, d) or acc_out is None:
                raise ValueError('axpy operand of shape %s for an accumulator of shape %s' % (tuple(ax.shape), (plan.n_rows, d)))
            epi.axpy_x, epi.axpy_alpha = keep_alive[-1].data_ptr(), float(alpha)
            if scale is not None:
                keep_alive.append(scale.reshape(1).to(torch.float32).contiguous())
                epi.axpy_scale = keep_alive[-1].data_ptr()
        if noise is not None and not torch.is_tensor(noise):      # rng.PhiloxNoise: computed in the epilogue
            want = (plan.n_rows, d) if noise_geom is None else (plan.n_rows, int(noise_geom[0]))
            if tuple(noise.shape) != want:
                raise ValueError('noise of shape %s for an output of shape %s' % (noise.shape, want))
            epi.noise, epi.philox, epi.philox_stream = None, noise.state.state.data_ptr(), int(noise.stream)
        else:
            epi.noise = _ptr(_f32c(noise)) if noise is not None else None
        epi.eps = float(eps)
        epi.acc_in = _ptr(acc_in)
        epi.acc_out = _ptr(acc_out)
        if sum_in:
            if len(sum_in) > MAX_SUM_IN or acc_out is None:
                raise ValueError('at most %d deferred layer tables, and an accumulator to add them to' % MAX_SUM_IN)
            epi.n_sum_in = len(sum_in)
            for j, t in enumerate(sum_in):
                if tuple(t.shape) != (plan.n_rows, d) or not t.is_contiguous() or t.dtype != torch.float32:
                    raise ValueError('deferred layer table %d: contiguous fp32 [%d, %d] expected' % (j, plan.n_rows, d))
                epi.sum_in[j] = t.data_ptr()
    lib = _lib.load()
    swept = plan.swept(d)
    lay = col = val = r_len = w_len = None
    if swept is not None:
        if view is not None:
            col, val, w_len = view.masked(which, d)   # (pack, val, w_steps) overrides
    elif scale_flags:
        raise ValueError('the factorized normalization (scale_flags) is the column-swept kernel\'s')
    else:
        lay = plan.packed(d)
        if view is not None:
            col, val, r_len, w_len = view.compact(which, d)
    prof, ev0, ev1 = _profile_this_launch(), None, None
    if prof:                     # (an event pair of its own per launch: sharing one event between back-to-back launches would
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # charge the HOST's enqueue gaps of a
        ev0.record()                                                                               # launch-bound step to the kernel)
    if STAMPS is not None:
        STAMPS.attach_next(swept if swept is not None else lay, d, acc_out is not None, want_y, _entry_frac(view), None, len(sum_in or ()))
    if swept is not None:       # output table fits the chip's LDS: column-swept kernel (spmm_swept.hip)
        rc = lib.sslrec_spmm_swept_f32(C.byref(swept.c_struct()), _ptr(col), _ptr(val), _ptr(w_len), x.data_ptr(), d,
                                       _ptr(y) if want_y else None,
                                       C.byref(epi) if epi is not None else None, _stream())
        _lib.check(rc, 'sslrec_spmm_swept_f32')
        if PROFILE is not None:
            if prof:
                ev1.record()
            PROFILE.append((ev0, ev1, swept, d, acc_out is not None, want_y, _entry_frac(view),
                            x_row_bits.max_rows if (x_row_bits is not None and epi is not None) else None, len(sum_in or ()),
                            bool(scale_flags & SCALE_PATTERN),
                            {'views': 1, 'perturbed': noise is not None, 'philox': noise is not None and not torch.is_tensor(noise), 'which': which}))
        return y if want_y else None
    if isinstance(lay, BundledLayout):      # narrow table beyond the swept layout: row-bundled kernel (spmm_bundle_kernel)
        # (col, val, b_steps, w_blocks) of a view: a compacted edge-dropped view brings all four, a re-valued one only the values
        rc = lib.sslrec_spmm_bundled_view_f32(C.byref(lay.c_struct()), _ptr(col), _ptr(val), _ptr(r_len), _ptr(w_len), x.data_ptr(), d,
                                              _ptr(y) if want_y else None, C.byref(epi) if epi is not None else None,
                                              _ptr(lay.partial_ws()), _stream())
        _lib.check(rc, 'sslrec_spmm_bundled_view_f32')
    else:
        rc = lib.sslrec_spmm_csr_f32(C.byref(lay.c_struct()), _ptr(col), _ptr(val), _ptr(r_len), _ptr(w_len),
                                     x.data_ptr(), d,
                                     _ptr(y) if want_y else None, C.byref(epi) if epi is not None else None,
                                     _ptr(lay.partial_ws()), _stream())
        _lib.check(rc, 'sslrec_spmm_csr_f32')
    if PROFILE is not None:
        if prof:
            ev1.record()
        PROFILE.append((ev0, ev1, lay, d, acc_out is not None, want_y, _entry_frac(view)))
    return y if want_y else None


def _entry_frac(view):
    """share of the matrix entries a launch reads (for the algorithmic-byte accounting of the measurement hook): 1 for
    the plain graph, the keep rate of a Philox edge-dropped view (its mask is never materialized)"""
    if isinstance(view, DroppedView) and view.keep is None:
        return float(view.philox[2])
    return 1.0


def _as_adj(adj):
    if isinstance(adj, (PropGraph, DroppedView, RevaluedView)):
        return adj
    return graph_of(adj)


# ----------------------------------------------------------------------------------------------
# Y = A X   (drop-in for torch.spmm in `_propagate`, reference lightgcn.py:28-29)
# ----------------------------------------------------------------------------------------------
class _SpmmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj):
        ctx.adj = adj
        return spmm_raw(adj, x, 'fwd')

    @staticmethod
    def backward(ctx, gy):
        return spmm_raw(ctx.adj, gy.contiguous(), 'bwd'), None


def _spmm_dim(adj, d):
    """embedding size the SpMM runs at: d itself when a kernel exists for it, else the next supported size (zero-padded)"""
    if d in SPMM_NARROW_DIMS:
        return d
    return _padded_dim(d, SPMM_DIMS)


def _padded_dim(d, supported):
    for s in supported:
        if d <= s:
            return s
    raise ValueError('embedding size %d exceeds the largest supported size %d' % (d, supported[-1]))


def _pad_cols(x, dp):
    """zero-pad the embedding dimension to a kernel-supported size (differentiable); zero columns
    change neither products, norms nor dot products, the result is sliced back by the caller"""
    d = x.shape[-1]
    return x if d == dp else torch.nn.functional.pad(x, (0, dp - d))


def spmm(adj, x):
    d = x.shape[1]
    adj = _as_adj(adj)
    dp = _spmm_dim(adj, d)
    y = _SpmmFn.apply(_pad_cols(x, dp), adj)
    return y if dp == d else y[:, :d]


# ----------------------------------------------------------------------------------------------
# fused L-layer propagation + layer SUM (+ optional per-layer perturbation)
#   S = E0 + sum_l E_l,  E_l = P_l(A E_{l-1})      (lightgcn.py:31-43 / simgcl.py:20-30)
# backward:  g_L = G,  g_{l-1} = G + A^T g_l,  dE0 = g_0   (perturbation has unit Jacobian a.e.)
# ----------------------------------------------------------------------------------------------
class RowBits:
    """Bitmap over the rows of a table: bit clear = the row is all zeros (sslrec_row_bits3).  `max_rows` = how many bits can be set."""

    def __init__(self, bits, n_rows, max_rows):
        self.bits, self.n_rows, self.max_rows = bits, int(n_rows), int(max_rows)

    @classmethod
    def from_indices(cls, n_rows, idx0, off0=0, idx1=None, off1=0, idx2=None, off2=0):
        if n_rows > (1 << 20):
            return None
        bits = torch.empty((n_rows + 31) // 32, dtype=torch.int32, device=idx0.device)
        _lib.check(_lib.load().sslrec_row_bits3(idx0.data_ptr(), int(off0), _ptr(idx1), int(off1), _ptr(idx2), int(off2), int(idx0.numel()),
                                                int(n_rows), bits.data_ptr(), _stream()), 'sslrec_row_bits3')
        k = int(idx0.numel()) * (1 + (idx1 is not None) + (idx2 is not None))
        return cls(bits, n_rows, min(n_rows, k))


SPARSE_GRAD = os.environ.get('SSLREC_SPARSE_GRAD', '1') != '0'      # the fused BPR backward tells the propagation which rows it wrote


def _tag_row_bits(t, rb):
    """remember on the tensor OBJECT which of its rows can be non-zero; valid while nobody writes to it (version counter)"""
    if rb is not None:
        t._sslrec_row_bits = (rb, t._version)


def _row_bits_of(t):
    tag = getattr(t, '_sslrec_row_bits', None)
    if tag is None or tag[1] != t._version or tag[0].n_rows != t.shape[0]:
        return None
    return tag[0]


class _PropagateSumFn(torch.autograd.Function):
    """outputs: (total, [reg], [layer 1 .. layer L]).  reg_weight (optional) adds `reg = reg_weight * sum(e0^2)` (reg_params,
    loss_utils.py:20-24) as a second output whose gradient 2 * reg_weight * g_reg * e0 is folded into the epilogue of the
    LAST backward product -- the step then needs neither a pass of its own over the table for it nor an elementwise add."""

    @staticmethod
    def forward(ctx, e0, adj, layer_num, noises, eps, keep_layers, noise_sumsq, noise_geom, reg_weight):
        ctx.adj, ctx.layer_num, ctx.reg_weight = adj, layer_num, reg_weight
        e0 = _f32c(e0)
        reg = ()
        if reg_weight is not None:
            lib = _lib.load()
            ws = _ticket_ws(e0.device, lib.sslrec_sumsq_ws_bytes(), 'sumsq')
            out = torch.empty(1, dtype=torch.float32, device=e0.device)
            _lib.check(lib.sslrec_sumsq_fwd_f32(e0.data_ptr(), e0.numel(), float(reg_weight), ws.data_ptr(), out.data_ptr(), _stream()),
                       'sslrec_sumsq_fwd_f32')
            reg = (out.reshape(()),)
            ctx.save_for_backward(e0)
        layers = [e0] if keep_layers else None
        ctx.row_scale = None
        if layer_num == 0:
            return (e0.clone(),) + reg
        total = torch.empty_like(e0)
        x = e0
        # deferred layer sum: the launches l < L write E_l only, the last one forms ((E0 + E1) + ...) + E_L -- the running sum's bits
        # without its 2 (L - 1) table-sized writes and L - 1 reads (column-swept layouts; L - 1 tables fit sslrec_epilogue_t.sum_in)
        plan_f = (adj.graph if isinstance(adj, (DroppedView, RevaluedView)) else adj).fwd
        lay_f = plan_f.swept(e0.shape[1]) if DEFERRED_SUM and 2 <= layer_num <= MAX_SUM_IN + 1 else None
        deferred = lay_f is not None and bool(_lib.load().sslrec_swept_deferred_sum_ok(C.byref(lay_f.c_struct())))
        # factorized normalization: launch 1 reads the values and writes the SCALED table r (.) E_1, the others add rows of the scaled
        # table (no value stream) and scale the row sum in their flush
        rsc = None if (deferred or keep_layers or noise_sumsq is not None or noise_geom is not None) else \
            _chain_scale(adj, e0.shape[1], layer_num, perturbed=noises is not None)
        ctx.row_scale = rsc
        mids = []
        for l in range(layer_num):
            last = (l == layer_num - 1)
            want_y = (not last) or keep_layers
            nz = None if noises is None else noises[l]
            nss = None if noise_sumsq is None else noise_sumsq[l]
            if deferred and not last:
                y = spmm_raw(adj, x, 'fwd', noise=nz, eps=eps, noise_sumsq=nss, noise_geom=noise_geom)
                mids.append(y)
            elif deferred:
                y = spmm_raw(adj, x, 'fwd', noise=nz, eps=eps, acc_in=e0, acc_out=total, want_y=want_y, sum_in=mids,
                             noise_sumsq=nss, noise_geom=noise_geom)
            else:
                y = spmm_raw(adj, x, 'fwd', noise=nz, eps=eps, acc_in=e0 if l == 0 else total, acc_out=total, want_y=want_y,
                             chained=l > 0, noise_sumsq=nss, noise_geom=noise_geom, row_scale=rsc,
                             scale_flags=0 if rsc is None else ((SCALE_PATTERN if l > 0 else 0) | (SCALE_Y if want_y else 0)))
            if keep_layers:
                layers.append(y)
            x = y
        ctx.mark_non_differentiable(*(layers[1:] if keep_layers else []))
        return (total,) + reg + (tuple(layers[1:]) if keep_layers else ())

    @staticmethod
    def backward(ctx, g_total, *rest):
        g_reg = rest[0] if ctx.reg_weight is not None else None
        e0 = ctx.saved_tensors[0] if ctx.reg_weight is not None else None
        if g_total is None:                 # only the regularizer was used
            return (None if g_reg is None else 2.0 * ctx.reg_weight * g_reg * e0,) + (None,) * 8
        sparse = _row_bits_of(g_total) if SPARSE_GRAD else None      # (before any copy: the tag lives on the tensor object autograd handed over)
        g_total = _f32c(g_total)
        if ctx.layer_num == 0:
            g = g_total if g_reg is None else g_total + 2.0 * ctx.reg_weight * g_reg * e0
            return (g,) + (None,) * 8
        g = g_total
        for l in range(ctx.layer_num):
            nxt = torch.empty_like(g_total)
            last = l == ctx.layer_num - 1
            rsc = ctx.row_scale      # (symmetric factors: A^T's row factor is A's)
            spmm_raw(ctx.adj, g, 'bwd', acc_in=g_total, acc_out=nxt, want_y=False, chained=l > 0,
                     axpy=(e0, 2.0 * ctx.reg_weight, g_reg) if (last and g_reg is not None) else None,
                     x_row_bits=sparse if l == 0 else None,      # A^T g: only the rows the loss wrote are gathered
                     row_scale=rsc, scale_flags=0 if rsc is None else ((SCALE_PATTERN if l > 0 else 0) | (0 if last else SCALE_ACC)))
            g = nxt
        return (g,) + (None,) * 8


def propagate_sum(adj, e0, layer_num, noises=None, eps=0.0, return_layers=False, noise_sumsq=None, noise_geom=None, reg_weight=None):
    """Sum over layers 0..L of the propagated embeddings, one fused kernel per layer.  With return_layers the per-layer
    tables come back too, for INSPECTION only: they are marked non-differentiable (the fused backward only propagates
    the gradient of the sum) -- a loss built on an individual layer must use ops.spmm per layer instead.
    reg_weight: also return `reg_weight * sum(e0^2)` (reg_params of the stacked table) -> (total, reg[, layers]); its gradient
    rides on the last backward product.  noise_sumsq (list of L [N] tensors) / noise_geom = (d_full, first column): the
    perturbation of a COLUMN SLICE of the tables (feature-sliced tables; see spmm_raw)."""
    d = e0.shape[1]
    adj = _as_adj(adj)
    dp = _spmm_dim(adj, d)
    if dp != d and noises is not None:
        if noise_geom is not None:
            raise ValueError('a column slice of %d columns is not a width of the kernels' % d)
        noises = [_pad_cols(n if torch.is_tensor(n) else n.materialize(), dp) for n in noises]
    out = _PropagateSumFn.apply(_pad_cols(e0, dp), adj, int(layer_num), noises, float(eps), bool(return_layers),
                                noise_sumsq, noise_geom, None if reg_weight is None else float(reg_weight))
    total, rest = out[0], list(out[1:])
    if dp != d:
        total = total[:, :d]
    res = [total]
    if reg_weight is not None:
        res.append(rest.pop(0))
    if return_layers:
        res.append([e0] + [l if dp == d else l[:, :d] for l in rest])
    return res[0] if len(res) == 1 else tuple(res)


def row_sumsq(x):
    """[N] sums of squares of the rows of a [N, w] table (a rank's share of EmbedPerturb's full-row noise norm, aug_utils.py:130)"""
    _need_gpu(x)
    x = _f32c(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sslrec_row_sumsq_f32(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream()), 'sslrec_row_sumsq_f32')
    return out


def philox_row_sumsq(token):
    """[N] squared norms of the rows of a computed noise table (rng.PhiloxNoise over the FULL [N, d] table): what a launch on a
    column slice of that table needs beside its own columns' draws"""
    n, d = token.shape
    out = torch.empty(n, dtype=torch.float32, device=token.state.state.device)
    _lib.check(_lib.load().sslrec_philox_row_sumsq(token.state.state.data_ptr(), int(token.stream), n, d, out.data_ptr(), _stream()),
               'sslrec_philox_row_sumsq')
    return out


class _PropagateSumViewsFn(torch.autograd.Function):
    """K views of the layer-summed propagation of the SAME table over the SAME (undropped) adjacency, differing only
    in their per-layer perturbation noise (None = clean view): SimGCL's three forwards (simgcl.py:29-31).  The first
    layer is one product A.E0 with K epilogues, and the backward pass -- the same linear map for every view -- runs once, on the
    SUM of the views' upstream gradients: K - 1 + (K - 1)(L - 1) SpMM launches fewer per step, same mathematics."""

    @staticmethod
    def forward(ctx, e0, graph, layer_num, noises_views, eps):
        e0 = _f32c(e0)
        K = len(noises_views)
        ctx.graph, ctx.layer_num, ctx.K = graph, layer_num, K
        n, d = e0.shape
        lay = graph.fwd.swept(d)
        totals = [torch.empty_like(e0) for _ in range(K)]
        xs = [torch.empty_like(e0) if layer_num > 1 else None for _ in range(K)]
        rsc = _chain_scale(graph, d, layer_num, perturbed=any(nz is not None for nz in noises_views))
        ctx.row_scale = rsc
        v = _lib.EpilogueViewsStruct()
        v.n_views, v.eps = K, float(eps)
        if rsc is not None:
            v.row_scale, v.scale_flags = rsc.data_ptr(), SCALE_Y
        keep_alive = []
        for k in range(K):
            nz = None if noises_views[k] is None else noises_views[k][0]
            if nz is not None and not torch.is_tensor(nz):            # rng.PhiloxNoise
                v.philox, v.philox_stream[k], v.philox_noise[k] = nz.state.state.data_ptr(), int(nz.stream), 1
                nz = None
            elif nz is not None:
                nz = _f32c(nz)
            keep_alive.append(nz)
            v.Y[k] = _ptr(xs[k]) or None
            v.noise[k] = _ptr(nz) or None
            v.acc_in[k] = e0.data_ptr()
            v.acc_out[k] = totals[k].data_ptr()
        ev0 = ev1 = None
        if _profile_this_launch():      # (the same sampling tick as every other SpMM launch: the sampled launch rotates through ALL of a step's)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = _lib.load().sslrec_spmm_swept_views_f32(C.byref(lay.c_struct()), e0.data_ptr(), d, C.byref(v), _stream())
        _lib.check(rc, 'sslrec_spmm_swept_views_f32')
        if PROFILE is not None:
            if ev1 is not None:
                ev1.record()
            PROFILE.append((ev0, ev1, lay, d, True, layer_num > 1, 1.0, None, 0, False,
                            {'views': K, 'perturbed': any(nz is not None for nz in noises_views),
                             'philox': any(v.philox_noise[k] for k in range(K)), 'which': 'fwd'}))
        for k in range(K):
            x = xs[k]
            for l in range(1, layer_num):
                last = (l == layer_num - 1)
                x = spmm_raw(graph, x, 'fwd', noise=None if noises_views[k] is None else noises_views[k][l], eps=eps,
                             acc_in=totals[k], acc_out=totals[k], want_y=not last, row_scale=rsc,
                             scale_flags=0 if rsc is None else (SCALE_PATTERN | (0 if last else SCALE_Y)))
        return tuple(totals)

    @staticmethod
    def backward(ctx, *g_totals):
        # The perturbation adds eps * sign(y) * noise / |noise| (aug_utils.py:125-132): its derivative with respect to y is the identity,
        # so every view's backward pass is the SAME linear map  g -> g + A^T (g + A^T (g + ...))  of its upstream gradient, and the sum
        # over the views of the maps is the map of the sum: L products for all K views instead of K (L - 1) + 1, and K - 1 table
        # additions instead of 2 (K - 1).  (SimGCL, K = 3, L = 2: 4 -> 2 backward products per step.)
        graph, L = ctx.graph, ctx.layer_num
        grads = [_f32c(g) for g in g_totals if g is not None]
        if not grads:
            return None, None, None, None, None
        G = grads[0]
        for i, g in enumerate(grads[1:]):
            G = torch.add(G, g) if i == 0 else G.add_(g)
        g = G
        rsc = ctx.row_scale
        for l in range(L):
            nxt = torch.empty_like(G)
            spmm_raw(graph, g, 'bwd', acc_in=G, acc_out=nxt, want_y=False, row_scale=rsc,
                     scale_flags=0 if rsc is None else ((SCALE_PATTERN if l > 0 else 0) | (0 if l == L - 1 else SCALE_ACC)))
            g = nxt
        return g, None, None, None, None


class _PropagateSumViewsLoopFn(torch.autograd.Function):
    """The same K views where the first layer's product cannot be shared (a table beyond the column-swept layout, a narrow or padded
    width, a column slice of the tables with its noise geometry): K separate forward chains, but still ONE backward chain on the
    summed upstream gradients (see _PropagateSumViewsFn.backward: the perturbation's derivative is the identity)."""

    @staticmethod
    def forward(ctx, e0, adj, layer_num, noises_views, eps, sumsq_views, noise_geom):
        e0 = _f32c(e0)
        ctx.adj, ctx.layer_num = adj, layer_num
        rsc = _chain_scale(adj, e0.shape[1], layer_num, perturbed=any(nz is not None for nz in noises_views)) \
            if (sumsq_views is None and noise_geom is None) else None
        ctx.row_scale = rsc
        totals = []
        for k, nzs in enumerate(noises_views):
            total = torch.empty_like(e0)
            x = e0
            for l in range(layer_num):
                last = (l == layer_num - 1)
                x = spmm_raw(adj, x, 'fwd', noise=None if nzs is None else nzs[l], eps=eps, acc_in=e0 if l == 0 else total, acc_out=total,
                             want_y=not last, noise_sumsq=None if (nzs is None or sumsq_views is None or sumsq_views[k] is None) else sumsq_views[k][l],
                             noise_geom=None if nzs is None else noise_geom, row_scale=rsc,
                             scale_flags=0 if rsc is None else ((SCALE_PATTERN if l > 0 else 0) | (0 if last else SCALE_Y)))
            totals.append(total)
        return tuple(totals)

    @staticmethod
    def backward(ctx, *g_totals):
        grads = [_f32c(g) for g in g_totals if g is not None]
        if not grads:
            return (None,) * 7
        G = grads[0]
        for i, g in enumerate(grads[1:]):
            G = torch.add(G, g) if i == 0 else G.add_(g)
        g = G
        rsc, L = ctx.row_scale, ctx.layer_num
        for l in range(L):
            nxt = torch.empty_like(G)
            spmm_raw(ctx.adj, g, 'bwd', acc_in=G, acc_out=nxt, want_y=False, row_scale=rsc,
                     scale_flags=0 if rsc is None else ((SCALE_PATTERN if l > 0 else 0) | (0 if l == L - 1 else SCALE_ACC)))
            g = nxt
        return (g,) + (None,) * 6


def propagate_sum_views(adj, e0, layer_num, noises_views, eps=0.0, noise_sumsq_views=None, noise_geom=None):
    """[propagate_sum(adj, e0, L, noises_k, eps) for noises_k in noises_views] -- views of the SAME table over the SAME adjacency that
    differ only in their perturbation noise (None = the clean view) -- with ONE backward chain for all views and, on the column-swept
    layout of the plain graph, the first layer's product shared too (K epilogues).  noise_sumsq_views (per view: L [N] tensors, or
    None) / noise_geom: the perturbation of a COLUMN SLICE of the tables (see propagate_sum)."""
    adj = _as_adj(adj)
    d = e0.shape[1]
    if layer_num < 1 or len(noises_views) < 2:
        return [propagate_sum(adj, e0, layer_num, nz, eps, noise_sumsq=None if noise_sumsq_views is None else noise_sumsq_views[k],
                              noise_geom=None if nz is None else noise_geom) for k, nz in enumerate(noises_views)]
    shared = (isinstance(adj, PropGraph) and d in SPMM_DIMS and len(noises_views) <= 4 and noise_sumsq_views is None and noise_geom is None
              and adj.bwd is not None and adj.fwd.swept(d) is not None)
    if shared:
        return list(_PropagateSumViewsFn.apply(e0, adj, int(layer_num), list(noises_views), float(eps)))
    dp = _spmm_dim(adj, d)
    if dp != d:
        if noise_geom is not None:
            raise ValueError('a column slice of %d columns is not a width of the kernels' % d)
        noises_views = [None if nzs is None else [_pad_cols(n if torch.is_tensor(n) else n.materialize(), dp) for n in nzs] for nzs in noises_views]
    outs = _PropagateSumViewsLoopFn.apply(_pad_cols(e0, dp), adj, int(layer_num), list(noises_views), float(eps), noise_sumsq_views, noise_geom)
    return [t if dp == d else t[:, :d] for t in outs]


# ----------------------------------------------------------------------------------------------
# BPR
# ----------------------------------------------------------------------------------------------
class _BprFn(torch.autograd.Function):
    """loss = sum_b f(<a,n> - <a,p>) over rows gathered from up to three tables."""

    @staticmethod
    def forward(ctx, ta, tp, tn, ia, ip, in_, variant, shared_pn, divisor=1.0):
        _need_gpu(ta, tp, tn)
        ta, tp, tn = _f32c(ta), _f32c(tp), _f32c(tn)
        ia, ip, in_ = _idx(ia), _idx(ip), _idx(in_)
        B = int(ia.numel()) if ia is not None else ta.shape[0]
        d = ta.shape[1]
        lib = _lib.load()
        ws = _ticket_ws(ta.device, lib.sslrec_bpr_ws_bytes(B), 'bpr')
        out = torch.empty(1, dtype=torch.float32, device=ta.device)
        rc = lib.sslrec_bpr_fwd_f32(ta.data_ptr(), _ptr(ia), tp.data_ptr(), _ptr(ip), tn.data_ptr(), _ptr(in_), B, d,
                                    variant, float(divisor), ws.data_ptr(), out.data_ptr(), _stream())
        _lib.check(rc, 'sslrec_bpr_fwd_f32')
        ctx.save_for_backward(ta, tp, tn, ia if ia is not None else torch.empty(0), ip if ip is not None else torch.empty(0),
                              in_ if in_ is not None else torch.empty(0))
        ctx.has_idx = (ia is not None, ip is not None, in_ is not None)
        ctx.meta = (B, d, variant, shared_pn, float(divisor))
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        ta, tp, tn, ia, ip, in_ = ctx.saved_tensors
        ia = ia if ctx.has_idx[0] else None
        ip = ip if ctx.has_idx[1] else None
        in_ = in_ if ctx.has_idx[2] else None
        B, d, variant, shared_pn, divisor = ctx.meta
        g = g.reshape(1).to(torch.float32).contiguous()
        dta = torch.zeros_like(ta) if ia is not None else torch.empty_like(ta)
        if shared_pn:                       # positives and negatives index the SAME table
            dtp = torch.zeros_like(tp)
            dtn = dtp
        else:
            dtp = torch.zeros_like(tp) if ip is not None else torch.empty_like(tp)
            dtn = torch.zeros_like(tn) if in_ is not None else torch.empty_like(tn)
        lib = _lib.load()
        ws, kept = _bpr_bwd_ws(ta.device, B, d)
        args = (ta.data_ptr(), _ptr(ia), tp.data_ptr(), _ptr(ip), tn.data_ptr(), _ptr(in_), B, d,
                variant, divisor, g.data_ptr(), dta.data_ptr(), dtp.data_ptr(), dtn.data_ptr(), ws.data_ptr())
        rc = lib.sslrec_bpr_bwd_kept_f32(*args, None, 0, _stream()) if kept else lib.sslrec_bpr_bwd_f32(*args, _stream())
        _check_kept(rc, 'sslrec_bpr_bwd_f32', ta.device, B, d)
        return dta, dtp, (None if shared_pn else dtn), None, None, None, None, None, None


def bpr_loss_and_grads(anc, pos, neg, variant=0, divisor=1.0):
    """cal_bpr_loss(anc, pos, neg) / divisor (loss_utils.py:7-10) and its gradients w.r.t. the three dense [B, d] operands in
    two launches, WITHOUT autograd: the building block of a hand-written backward pass (sslrec_amd/feature_shard.py's captured
    step).  Returns (loss [1], d_anc, d_pos, d_neg)."""
    _need_gpu(anc, pos, neg)
    anc, pos, neg = _f32c(anc.detach()), _f32c(pos.detach()), _f32c(neg.detach())
    B, d = anc.shape
    lib = _lib.load()
    dev = anc.device
    ws = _ticket_ws(dev, lib.sslrec_bpr_ws_bytes(B), 'bpr')
    out = torch.empty(1, dtype=torch.float32, device=dev)
    rc = lib.sslrec_bpr_fwd_f32(anc.data_ptr(), None, pos.data_ptr(), None, neg.data_ptr(), None, B, d, int(variant),
                                float(divisor), ws.data_ptr(), out.data_ptr(), _stream())
    _lib.check(rc, 'sslrec_bpr_fwd_f32')
    one = torch.ones(1, dtype=torch.float32, device=dev)
    da, dp, dn = torch.empty_like(anc), torch.empty_like(pos), torch.empty_like(neg)
    ws2 = torch.empty(lib.sslrec_bpr_bwd_ws_bytes(B, d) // 4 + 1, dtype=torch.float32, device=dev)
    rc = lib.sslrec_bpr_bwd_f32(anc.data_ptr(), None, pos.data_ptr(), None, neg.data_ptr(), None, B, d, int(variant),
                                float(divisor), one.data_ptr(), da.data_ptr(), dp.data_ptr(), dn.data_ptr(), ws2.data_ptr(), _stream())
    _lib.check(rc, 'sslrec_bpr_bwd_f32')
    return out, da, dp, dn


def bpr_loss(anc, pos, neg, variant=0, divisor=1.0):
    """Dense drop-in for cal_bpr_loss(anc[B,d], pos[B,d], neg[B,d]) (loss_utils.py:7-10); returns the SUM, divided by
    `divisor` inside the kernel (pass the batch size to fold the reference's `/ ancs.shape[0]`)."""
    return _BprFn.apply(anc, pos, neg, None, None, None, int(variant), False, float(divisor))


class _BprStackedFn(torch.autograd.Function):
    """BPR over ONE stacked table [users; items]: anchors index its first n_user rows, positives / negatives the rows
    after them (the item part is addressed through an offset base pointer -- no index arithmetic); one gradient buffer.
    With `add` (a 0-d tensor, e.g. the regularizer term) the SAME launch also returns total = bpr + add (lightgcn.py:54's
    `bpr_loss + reg_loss`): outputs (total, bpr), the gradient of `total` flows to the table and, unchanged, to `add`."""

    @staticmethod
    def forward(ctx, table, n_user, ancs, poss, negs, variant, divisor, add):
        _need_gpu(table)
        table = _f32c(table)
        ia, ip, in_ = _idx(ancs), _idx(poss), _idx(negs)
        B, d = int(ia.numel()), table.shape[1]
        lib = _lib.load()
        ws = _ticket_ws(table.device, lib.sslrec_bpr_ws_bytes(B), 'bpr')
        p = table.data_ptr()
        pi = p + int(n_user) * d * 4
        ctx.save_for_backward(table, ia, ip, in_)
        ctx.meta = (B, d, variant, int(n_user), float(divisor))
        ctx.two = add is not None
        ctx.set_materialize_grads(False)
        if add is None:
            out = torch.empty(1, dtype=torch.float32, device=table.device)
            rc = lib.sslrec_bpr_fwd_f32(p, ia.data_ptr(), pi, ip.data_ptr(), pi, in_.data_ptr(), B, d, variant, float(divisor),
                                        ws.data_ptr(), out.data_ptr(), _stream())
            _lib.check(rc, 'sslrec_bpr_fwd_f32')
            return out.reshape(())
        _need_gpu(add)
        addc = add.detach().reshape(1).to(torch.float32).contiguous()
        out = torch.empty(2, dtype=torch.float32, device=table.device)
        rc = lib.sslrec_bpr_fwd_total_f32(p, ia.data_ptr(), pi, ip.data_ptr(), pi, in_.data_ptr(), B, d, variant, float(divisor),
                                          addc.data_ptr(), ws.data_ptr(), out.data_ptr(), _stream())
        _lib.check(rc, 'sslrec_bpr_fwd_total_f32')
        return out[1].reshape(()), out[0].reshape(())

    @staticmethod
    def backward(ctx, *grads):
        table, ia, ip, in_ = ctx.saved_tensors
        B, d, variant, n_user, divisor = ctx.meta
        if ctx.two:                  # outputs (total, bpr): total's gradient goes to the table AND to `add`; bpr's (if anybody
            g_total, g_bpr = grads   # differentiates the logged part too) to the table only
            g_add = g_total
            g = g_total if g_bpr is None else (g_bpr if g_total is None else g_total + g_bpr)
        else:
            g, g_add = grads[0], None
        if g is None:
            return None, None, None, None, None, None, None, g_add
        g = g.reshape(1).to(torch.float32).contiguous()
        lib = _lib.load()
        ws, kept = _bpr_bwd_ws(table.device, B, d)
        fused_zero = kept and B > 0 and table.numel() % 4 == 0      # the staging launch zeroes the gradient table itself: no fill launch
        grad = torch.empty_like(table) if fused_zero else torch.zeros_like(table)
        p, q = table.data_ptr(), grad.data_ptr()
        pi, qi = p + n_user * d * 4, q + n_user * d * 4
        args = (p, ia.data_ptr(), pi, ip.data_ptr(), pi, in_.data_ptr(), B, d, variant, divisor, g.data_ptr(), q, qi, qi, ws.data_ptr())
        if kept:
            rc = lib.sslrec_bpr_bwd_kept_f32(*args, q if fused_zero else None, grad.numel() if fused_zero else 0, _stream())
        else:
            rc = lib.sslrec_bpr_bwd_f32(*args, _stream())
        _check_kept(rc, 'sslrec_bpr_bwd_f32', table.device, B, d)
        if SPARSE_GRAD:      # rows ancs / n_user + poss / n_user + negs are the only ones written
            _tag_row_bits(grad, RowBits.from_indices(table.shape[0], ia, 0, ip, n_user, in_, n_user))
        return grad, None, None, None, None, None, None, g_add


def bpr_loss_stacked(table, n_user, ancs, poss, negs, variant=0, divisor=1.0, add=None):
    """Fused gather + BPR on the stacked [users; items] table the propagation produces (no slicing,
    one [N, d] gradient buffer): rows ancs / n_user + poss / n_user + negs (lightgcn.py:49-52).
    add (0-d tensor): returns (bpr + add, bpr) from the same launch."""
    return _BprStackedFn.apply(table, int(n_user), ancs, poss, negs, int(variant), float(divisor), add)


def bpr_loss_gathered(user_table, item_table, ancs, poss, negs, variant=0, divisor=1.0):
    """Fused gather + BPR: rows user_table[ancs], item_table[poss], item_table[negs]
    (lightgcn.py:49-52) without materializing the three [B,d] gathers."""
    return _BprFn.apply(user_table, item_table, item_table, ancs, poss, negs, int(variant), True, float(divisor))


# ----------------------------------------------------------------------------------------------
# InfoNCE
# ----------------------------------------------------------------------------------------------
def _infonce_event():
    if PROFILE_INFONCE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _infonce_record(ev0, kind, B, M, d, variant):
    if ev0 is not None and PROFILE_INFONCE is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        PROFILE_INFONCE.append((ev0, ev1, kind, B, M, d, variant))


def infonce_issued_flops(kind, B, M, d, variant):
    """matrix-core work ONE call issues, as (flops, 'bf16' | 'fp32'): a B x M x d product is 2BMd flops in exact-fp32 MFMA and
    (planes-dependent) 3 or 6 bf16 MFMA terms of 2BMd each in the split-precision modes (csrc/infonce_x3.inc).  Forward = the
    scores (+ the anchor-gradient product under SSLREC_INFONCE_FWD_W); backward = scores + the `all`-gradient product (+ scores and
    the anchor-gradient product again without the flag)."""
    code = (variant >> 8) & 0xFF
    v1 = (variant & 0xFF) != 0
    if code == 0:      # the library's defaults (csrc/infonce.hip, inf_precision): h3 -- on the un-normalized variant unless SSLREC_INFONCE_V1_DEFAULT=x6
        v1_default = 'x6' if (os.environ.get('SSLREC_INFONCE_V1_DEFAULT') or 'h3')[0] == 'x' else 'h3'
        code = INFONCE_PRECISIONS.get(os.environ.get('SSLREC_INFONCE_PRECISION') or (v1_default if v1 else 'h3'), 1)
    if v1 and code not in (2, 7):      # un-normalized rows: x6, exact fp32, or h3 with device-chosen scales; the other modes run x6
        code = 1
    fwd_w = bool(variant & INFONCE_FWD_W_BIT) and not (code == 2 and d == 128)
    # terms per (score product, anchor-gradient product, all-gradient product)
    sc, wa, da = {1: (6, 6, 6), 2: (1, 1, 1), 3: (3, 6, 6), 4: (3, 3, 3), 5: (6, 3, 3), 6: (6, 6, 3), 7: (3, 3, 3)}[code]
    unit = 2.0 * B * M * d
    pre = 1 if (v1 and code == 7 and (kind == 'fwd')) else 0      # h3 on the un-normalized variant: the row-max pre-pass, one term on the high planes
    if kind == 'fwd':
        f = sc + (wa if fwd_w else 0) + pre
    else:
        f = sc + da + (0 if fwd_w else sc + wa)
    return f * unit, ('fp32' if code == 2 else ('fp16' if code == 7 else 'bf16'))


class _InfoNceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t1, t2, all_, i1, i2, temp, variant, t2_is_all, will_differentiate=True):
        _need_gpu(t1, t2, all_)
        t1, t2, all_ = _f32c(t1), _f32c(t2), _f32c(all_)
        i1, i2 = _idx(i1), _idx(i2)
        B = int(i1.numel()) if i1 is not None else t1.shape[0]
        M, d = all_.shape
        if d not in INFONCE_DIMS:
            raise ValueError('embedding size %d not supported by the HIP InfoNCE (supported: %s)' % (d, INFONCE_DIMS))
        lib = _lib.load()
        if INFONCE_FWD_W and will_differentiate and any(ctx.needs_input_grad[:3]):
            variant |= INFONCE_FWD_W_BIT
        ws = torch.empty(lib.sslrec_infonce_ws_bytes(B, M, d) // 4, dtype=torch.float32, device=t1.device)
        out = torch.empty(1, dtype=torch.float32, device=t1.device)
        ev = _infonce_event()
        rc = lib.sslrec_infonce_fwd_f32(t1.data_ptr(), _ptr(i1), t2.data_ptr(), _ptr(i2), B, all_.data_ptr(), M, d,
                                        float(temp), variant, ws.data_ptr(), out.data_ptr(), _stream())
        _lib.check(rc, 'sslrec_infonce_fwd_f32')
        _infonce_record(ev, 'fwd', B, M, d, variant)
        ctx.save_for_backward(t1, t2, all_, i1 if i1 is not None else torch.empty(0),
                              i2 if i2 is not None else torch.empty(0), ws)
        ctx.has_idx = (i1 is not None, i2 is not None)
        ctx.meta = (B, M, d, float(temp), variant, t2_is_all)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        t1, t2, all_, i1, i2, ws = ctx.saved_tensors
        i1 = i1 if ctx.has_idx[0] else None
        i2 = i2 if ctx.has_idx[1] else None
        B, M, d, temp, variant, t2_is_all = ctx.meta
        g = g.reshape(1).to(torch.float32).contiguous()
        dev = t1.device
        lib = _lib.load()
        ev = _infonce_event()
        if i1 is not None and i2 is not None and 2 * B <= 16384:
            # both roles gathered (simgcl.py:49 / sgl.py:57-59): backward + one deterministic scatter for both in one call
            de = torch.empty((2 * B, d), dtype=torch.float32, device=dev)
            dall = torch.empty((M, d), dtype=torch.float32, device=dev)
            dt1 = torch.zeros_like(t1)
            dt2 = dall if t2_is_all else torch.zeros_like(t2)
            sws = torch.empty(lib.sslrec_scatter_ws_bytes(2 * B) // 4 + 1, dtype=torch.float32, device=dev)
            rc = lib.sslrec_infonce_bwd_scatter_f32(t1.data_ptr(), i1.data_ptr(), t2.data_ptr(), i2.data_ptr(), B, all_.data_ptr(), M, d,
                                                    temp, variant, ws.data_ptr(), g.data_ptr(), de.data_ptr(), dt1.data_ptr(),
                                                    dt2.data_ptr(), dall.data_ptr(), sws.data_ptr(), _stream())
            _lib.check(rc, 'sslrec_infonce_bwd_scatter_f32')
            _infonce_record(ev, 'bwd', B, M, d, variant)
            return (dt1, None, dall, None, None, None, None, None, None) if t2_is_all else (dt1, dt2, dall, None, None, None, None, None, None)
        de1 = torch.empty((B, d), dtype=torch.float32, device=dev)
        de2 = torch.empty((B, d), dtype=torch.float32, device=dev)
        dall = torch.empty((M, d), dtype=torch.float32, device=dev)
        rc = lib.sslrec_infonce_bwd_f32(t1.data_ptr(), _ptr(i1), t2.data_ptr(), _ptr(i2), B, all_.data_ptr(), M, d,
                                        temp, variant, ws.data_ptr(), g.data_ptr(), de1.data_ptr(), de2.data_ptr(),
                                        dall.data_ptr(), _stream())
        _lib.check(rc, 'sslrec_infonce_bwd_f32')
        _infonce_record(ev, 'bwd', B, M, d, variant)

        def scatter(src, idx, dst):
            sws = torch.empty(lib.sslrec_scatter_ws_bytes(B) // 4 + 1, dtype=torch.float32, device=src.device)
            rc2 = lib.sslrec_scatter_add_rows_f32(src.data_ptr(), idx.data_ptr(), B, d, dst.data_ptr(), sws.data_ptr(), _stream())
            _lib.check(rc2, 'sslrec_scatter_add_rows_f32')

        if i1 is not None:
            dt1 = torch.zeros_like(t1)
            scatter(de1, i1, dt1)
        else:
            dt1 = de1
        if t2_is_all:                       # e2 rows are gathered from the same table as `all`
            if i2 is not None:
                scatter(de2, i2, dall)
            else:
                dall += de2
            return dt1, None, dall, None, None, None, None, None, None
        if i2 is not None:
            dt2 = torch.zeros_like(t2)
            scatter(de2, i2, dt2)
        else:
            dt2 = de2
        return dt1, dt2, dall, None, None, None, None, None, None


def infonce_loss(embeds1, embeds2, all_embeds2, temp=1.0, variant=0, precision=None):
    """Dense drop-in for cal_infonce_loss(embeds1[B,d], embeds2[B,d], all_embeds2[M,d], temp)
    (loss_utils.py:30-39); returns the SUM over the batch.  `precision`: 'x6' | 'fp32' | 'x6a' | 'x63' | 'x36' | 'x3' | None (default)."""
    dp = _padded_dim(embeds1.shape[1], INFONCE_DIMS)
    return _InfoNceFn.apply(_pad_cols(embeds1, dp), _pad_cols(embeds2, dp), _pad_cols(all_embeds2, dp), None, None,
                            float(temp), _variant_code(variant, precision), False, torch.is_grad_enabled())


def infonce_loss_gathered(table1, table2, idx, temp=1.0, variant=0, precision=None):
    """Fused form of cal_infonce_loss(table1[idx], table2[idx], table2, temp) -- the call shape of
    simgcl.py:49 and sgl.py:57-59 -- without materializing the gathers."""
    dp = _padded_dim(table1.shape[1], INFONCE_DIMS)
    if dp != table1.shape[1]:
        table1, table2 = _pad_cols(table1, dp), _pad_cols(table2, dp)
    # (inside Function.forward grad mode is always off and needs_input_grad ignores no_grad(): the caller's grad mode is passed in)
    return _InfoNceFn.apply(table1, table2, table2, idx, idx, float(temp), _variant_code(variant, precision), True, torch.is_grad_enabled())


class _InfoNceTwoSidedFn(torch.autograd.Function):
    """cal_infonce_loss(U1[iu], U2[iu], U2, temp) + cal_infonce_loss(I1[ii], I2[ii], I2, temp) (simgcl.py:49, sgl.py:57-59) on the STACKED
    tables s = [U; I] that the propagation returns, as ONE autograd node: the gradients of both terms are written straight into one
    [N, d] buffer per view (the `all`-side gradient covers every row of view 2's buffer, so only view 1's is zero-filled).  Going
    through the split tables instead costs, per view and step, two slice-backward nodes (a table-sized zero fill + copy each) and
    the add that joins them -- ~150 us of stock elementwise launches per SimGCL step at amazon-book size (profiles/r04/cfg3_kernel_stats.csv)."""

    @staticmethod
    def forward(ctx, s1, s2, n_user, iu, ii, temp, variant, will_differentiate):
        _need_gpu(s1, s2)
        s1, s2 = _f32c(s1), _f32c(s2)
        iu, ii = _idx(iu), _idx(ii)
        N, d = s1.shape
        lib = _lib.load()
        if INFONCE_FWD_W and will_differentiate and any(ctx.needs_input_grad[:2]):
            variant |= INFONCE_FWD_W_BIT
        sides = [(0, int(n_user), iu), (int(n_user), N - int(n_user), ii)]
        outs = torch.empty(2, dtype=torch.float32, device=s1.device)
        wss = []
        for k, (row0, M, idx) in enumerate(sides):
            B = int(idx.numel())
            off = row0 * d * 4
            ws = torch.empty(lib.sslrec_infonce_ws_bytes(B, M, d) // 4, dtype=torch.float32, device=s1.device)
            ev = _infonce_event()
            rc = lib.sslrec_infonce_fwd_f32(s1.data_ptr() + off, idx.data_ptr(), s2.data_ptr() + off, idx.data_ptr(), B, s2.data_ptr() + off, M, d,
                                            float(temp), variant, ws.data_ptr(), outs.data_ptr() + 4 * k, _stream())
            _lib.check(rc, 'sslrec_infonce_fwd_f32')
            _infonce_record(ev, 'fwd', B, M, d, variant)
            wss.append(ws)
        ctx.save_for_backward(s1, s2, iu, ii, *wss)
        ctx.meta = (int(n_user), float(temp), variant)
        return outs[0] + outs[1]

    @staticmethod
    def backward(ctx, g):
        s1, s2, iu, ii, ws_u, ws_i = ctx.saved_tensors
        n_user, temp, variant = ctx.meta
        N, d = s1.shape
        g = g.reshape(1).to(torch.float32).contiguous()
        lib = _lib.load()
        g1 = torch.zeros_like(s1)             # the anchors' rows are scattered into it
        g2 = torch.empty_like(s2)             # every row is written: it is the `all` operand of one of the two terms
        for row0, M, idx, ws in ((0, n_user, iu, ws_u), (n_user, N - n_user, ii, ws_i)):
            B = int(idx.numel())
            off = row0 * d * 4
            de = torch.empty((2 * B, d), dtype=torch.float32, device=s1.device)
            sws = torch.empty(lib.sslrec_scatter_ws_bytes(2 * B) // 4 + 1, dtype=torch.float32, device=s1.device)
            ev = _infonce_event()
            rc = lib.sslrec_infonce_bwd_scatter_f32(s1.data_ptr() + off, idx.data_ptr(), s2.data_ptr() + off, idx.data_ptr(), B, s2.data_ptr() + off, M, d,
                                                    temp, variant, ws.data_ptr(), g.data_ptr(), de.data_ptr(), g1.data_ptr() + off,
                                                    g2.data_ptr() + off, g2.data_ptr() + off, sws.data_ptr(), _stream())
            _lib.check(rc, 'sslrec_infonce_bwd_scatter_f32')
            _infonce_record(ev, 'bwd', B, M, d, variant)
        return g1, g2, None, None, None, None, None, None


def infonce_loss_two_sided(stacked1, stacked2, n_user, user_idx, item_idx, temp=1.0, variant=0, precision=None):
    """cal_infonce_loss(U1[user_idx], U2[user_idx], U2, temp) + cal_infonce_loss(I1[item_idx], I2[item_idx], I2, temp) with
    [U_k; I_k] = stacked_k -- the contrastive term of SimGCL (simgcl.py:49) and, with item_idx = [poss; negs], of SGL (sgl.py:57-59) --
    on the propagation's stacked tables, one autograd node (see _InfoNceTwoSidedFn).  Embedding sizes without a kernel width and
    batches beyond the scatter table take the two gathered calls on the split tables."""
    d = stacked1.shape[1]
    if d not in INFONCE_DIMS or 2 * max(int(user_idx.numel()), int(item_idx.numel())) > 16384 or not stacked1.is_cuda:
        return infonce_loss_gathered(stacked1[:n_user], stacked2[:n_user], user_idx, temp, variant, precision) + \
            infonce_loss_gathered(stacked1[n_user:], stacked2[n_user:], item_idx, temp, variant, precision)
    return _InfoNceTwoSidedFn.apply(stacked1, stacked2, int(n_user), user_idx, item_idx, float(temp), _variant_code(variant, precision),
                                    torch.is_grad_enabled())


# ----------------------------------------------------------------------------------------------
# the stacked parameter table [user_embeds; item_embeds] without the per-forward concatenation (reference lightgcn.py:34)
# ----------------------------------------------------------------------------------------------
def stacked_alias(u, i):
    """the [U + I, d] table whose first U rows ARE `u` and whose last I rows ARE `i`, when the two tensors are adjacent row ranges of one
    contiguous fp32 buffer (how GraphCF allocates its two parameters); None otherwise.  No copy, detached from autograd."""
    if (u.dim() != 2 or i.dim() != 2 or u.shape[1] != i.shape[1] or u.dtype != torch.float32 or i.dtype != torch.float32 or u.device != i.device
            or not u.is_contiguous() or not i.is_contiguous()):
        return None
    if u.data_ptr() + u.numel() * 4 != i.data_ptr():
        return None
    su, si = u.untyped_storage(), i.untyped_storage()
    if su.data_ptr() != si.data_ptr() or (u.storage_offset() + u.numel() + i.numel()) * 4 > su.nbytes():
        return None
    return torch.empty(0, dtype=torch.float32, device=u.device).set_(su, u.storage_offset(), (u.shape[0] + i.shape[0], u.shape[1]), (u.shape[1], 1))


class _StackParamsFn(torch.autograd.Function):
    """cat([u, i]) for two tensors that already lie behind each other in memory: forward is an alias of their buffer, backward hands
    each its row range of the incoming gradient (views: AccumulateGrad keeps them, nothing is copied)"""

    @staticmethod
    def forward(ctx, u, i):
        ctx.n_u = u.shape[0]
        # The alias has a version counter of its own, so autograd cannot see through it that a tensor saved downstream (the table itself:
        # e0 of the regularizer / of the fused steps) changed when a PARAMETER was written in place between forward and backward -- the
        # reference's t.concat made a copy and was immune.  Saving the two parameters here makes autograd check THEIR counters when this
        # node runs backward (every consumer of the alias backpropagates through it): such a write raises instead of giving wrong numbers.
        ctx.save_for_backward(u, i)
        return stacked_alias(u, i)

    @staticmethod
    def backward(ctx, g):
        ctx.saved_tensors          # raises "modified by an inplace operation" when a parameter was written since the forward
        return g[:ctx.n_u], g[ctx.n_u:]


def stack_params(u, i):
    """[u; i] joined to autograd: without a copy when the two are adjacent row ranges of one buffer, else torch.cat"""
    if stacked_alias(u, i) is None:
        return torch.cat([u, i], dim=0)
    return _StackParamsFn.apply(u, i)


# ----------------------------------------------------------------------------------------------
# SimGCL / SGL: the whole training step as ONE autograd node with a hand-written backward
# ----------------------------------------------------------------------------------------------
# SSLREC_ONE_NODE_STEP=0: the models compose the step from the separate autograd nodes above (rounds 1-4; same kernels, plus the stock
# elementwise launches autograd needs between them: table-sized gradient additions, zero fills, scalar arithmetic)
ONE_NODE_STEP = os.environ.get('SSLREC_ONE_NODE_STEP', '1') != '0'


def _backward_chain(adj, G, L, d, axpy=None, sparse=None, sum_in_last=None):
    """g_L = G, g_{l-1} = G + A^T g_l; returns g_0 (the backward pass of the layer-summed propagation over `adj`).  axpy: fused into
    the last product (the regularizer's gradient); sparse: RowBits of G's non-zero rows (first product); sum_in_last: tables added by
    the last product's flush (other chains' results), or None"""
    rsc = _chain_scale(adj, d, L)
    g = G
    for l in range(L):
        last = l == L - 1
        nxt = torch.empty_like(G)
        spmm_raw(adj, g, 'bwd', acc_in=G, acc_out=nxt, want_y=False, axpy=axpy if last else None, x_row_bits=sparse if l == 0 else None,
                 sum_in=sum_in_last if last else None, row_scale=rsc,
                 scale_flags=0 if rsc is None else ((SCALE_PATTERN if l > 0 else 0) | (0 if last else SCALE_ACC)))
        g = nxt
    return g


class _ContrastiveStepFn(torch.autograd.Function):
    """loss = bpr(view3) / B + reg_weight |E0|^2 + cl_weight / B * (infonce(view1_u[a], view2_u[a], view2_u) + infonce(view1_i[p], view2_i[p],
    view2_i)) of SimGCL (simgcl.py:39-55: views = two perturbed propagations + a clean one over ONE adjacency) and SGL (sgl.py:45-65: two
    edge-dropped adjacencies + the clean one) as ONE autograd node on the two parameter tables.  Forward: the fused kernels of the
    separate nodes, no graph in between; the four scalars come out of one launch (sslrec_weighted_sum4_f32).  Backward, by hand:
      * SimGCL: all three views share the backward map (the perturbation's derivative is the identity), so the InfoNCE `all` gradients,
        the scattered anchor rows and the BPR rows are written into ONE table G -- no zero fill, no table addition -- and one chain of L
        products (the regularizer's gradient in the last flush) gives dE0;
      * SGL: three chains over three adjacencies; the two sparse upstream tables (anchor rows, BPR rows) tell their first product which
        rows are non-zero; the last product of the third chain adds the other two results in its flush (or one sslrec_add_tables launch
        where the layout's build has no room for that).
    Returns (loss, bpr, cl_weight * cl / B, reg) -- only `loss` carries gradient."""

    @staticmethod
    def forward(ctx, user_embeds, item_embeds, spec):
        _need_gpu(user_embeds, item_embeds)
        lib = _lib.load()
        e0 = stacked_alias(user_embeds, item_embeds)          # (GraphCF's two parameters share one buffer: no concatenation)
        if e0 is None:
            e0 = torch.cat([_f32c(user_embeds), _f32c(item_embeds)])
        n_user, (N, d), L = int(user_embeds.shape[0]), e0.shape, int(spec['layer_num'])
        dev = e0.device
        if spec['kind'] == 'simgcl':
            v1, v2, v3 = propagate_sum_views(spec['adj'], e0, L, [spec['noises'][0], spec['noises'][1], None], spec['eps'])
            adjs = (spec['adj'],) * 3
        else:
            adjs = tuple(spec['adjs'])
            v1, v2, v3 = (propagate_sum(a, e0, L) for a in adjs)
        ancs, poss, negs, items_cl = (_idx(spec[k]) for k in ('ancs', 'poss', 'negs', 'items_cl'))
        B = int(ancs.numel())
        # BPR on the clean view (lightgcn.py:49-52), already divided by B
        parts = torch.empty(4, dtype=torch.float32, device=dev)      # bpr, infonce users, infonce items, reg
        p3 = v3.data_ptr()
        p3i = p3 + n_user * d * 4
        ws_b = _ticket_ws(dev, lib.sslrec_bpr_ws_bytes(B), 'bpr')
        _lib.check(lib.sslrec_bpr_fwd_f32(p3, ancs.data_ptr(), p3i, poss.data_ptr(), p3i, negs.data_ptr(), B, d, 0, float(B), ws_b.data_ptr(),
                                          parts.data_ptr(), _stream()), 'sslrec_bpr_fwd_f32')
        variant = _variant_code(0, spec.get('precision')) | (INFONCE_FWD_W_BIT if INFONCE_FWD_W else 0)
        sides = [(0, n_user, ancs), (n_user, N - n_user, items_cl)]
        wss = []
        for k, (row0, M, idx) in enumerate(sides):
            Bk, off = int(idx.numel()), row0 * d * 4
            ws = torch.empty(lib.sslrec_infonce_ws_bytes(Bk, M, d) // 4, dtype=torch.float32, device=dev)
            ev = _infonce_event()
            _lib.check(lib.sslrec_infonce_fwd_f32(v1.data_ptr() + off, idx.data_ptr(), v2.data_ptr() + off, idx.data_ptr(), Bk, v2.data_ptr() + off,
                                                  M, d, float(spec['temp']), variant, ws.data_ptr(), parts.data_ptr() + 4 * (1 + k), _stream()),
                       'sslrec_infonce_fwd_f32')
            _infonce_record(ev, 'fwd', Bk, M, d, variant)
            wss.append(ws)
        ws_r = _ticket_ws(dev, lib.sslrec_sumsq_ws_bytes(), 'sumsq')
        _lib.check(lib.sslrec_sumsq_fwd_f32(e0.data_ptr(), e0.numel(), float(spec['reg_weight']), ws_r.data_ptr(), parts.data_ptr() + 12, _stream()),
                   'sslrec_sumsq_fwd_f32')
        out = torch.empty(6, dtype=torch.float32, device=dev)
        w_cl = float(spec['cl_weight']) / B
        pp = parts.data_ptr()
        _lib.check(lib.sslrec_weighted_sum4_f32(pp, 1.0, pp + 4, w_cl, pp + 8, w_cl, pp + 12, 1.0, out.data_ptr(), _stream()), 'sslrec_weighted_sum4_f32')
        # (the two parameters ride along so that an in-place write to them between forward and backward is detected: e0 may be an alias
        # of their buffer with a version counter of its own, see _StackParamsFn)
        ctx.save_for_backward(e0, v1, v2, v3, ancs, poss, negs, items_cl, *wss, user_embeds, item_embeds)
        ctx.meta = (spec['kind'], adjs, L, n_user, float(spec['temp']), variant, w_cl, float(spec['reg_weight']))
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[5], out[4]

    @staticmethod
    def backward(ctx, g_loss, g_bpr, g_cl, g_reg):
        e0, v1, v2, v3, ancs, poss, negs, items_cl, ws_u, ws_i, _, _ = ctx.saved_tensors
        kind, adjs, L, n_user, temp, variant, w_cl, reg_weight = ctx.meta
        if g_loss is None:
            return None, None, None
        if g_bpr is not None or g_cl is not None or g_reg is not None:
            raise RuntimeError('only the total loss of the fused contrastive step is differentiable (its parts are returned for logging)')
        lib = _lib.load()
        N, d = e0.shape
        dev = e0.device
        B = int(ancs.numel())
        g = g_loss.reshape(1).to(torch.float32).contiguous()
        cs = torch.empty(2, dtype=torch.float32, device=dev)          # (g * cl_weight / B, g)
        _lib.check(lib.sslrec_scalar_scale2_f32(g.data_ptr(), w_cl, 1.0, cs.data_ptr(), _stream()), 'sslrec_scalar_scale2_f32')
        c_cl, c_one = cs.data_ptr(), cs.data_ptr() + 4
        simgcl = kind == 'simgcl'
        # upstream tables of the three views: SimGCL -> one buffer for all of them
        G2 = torch.empty_like(e0)                       # every row is written: `all` operand of one of the two terms
        G1 = G2 if simgcl else torch.zeros_like(e0)     # anchor rows of view 1
        for row0, M, idx, ws in ((0, n_user, ancs, ws_u), (n_user, N - n_user, items_cl, ws_i)):
            Bk, off = int(idx.numel()), row0 * d * 4
            de = torch.empty((2 * Bk, d), dtype=torch.float32, device=dev)
            sws = torch.empty(lib.sslrec_scatter_ws_bytes(2 * Bk) // 4 + 1, dtype=torch.float32, device=dev)
            ev = _infonce_event()
            _lib.check(lib.sslrec_infonce_bwd_scatter_f32(v1.data_ptr() + off, idx.data_ptr(), v2.data_ptr() + off, idx.data_ptr(), Bk,
                                                          v2.data_ptr() + off, M, d, temp, variant, ws.data_ptr(), c_cl, de.data_ptr(),
                                                          G1.data_ptr() + off, G2.data_ptr() + off, G2.data_ptr() + off, sws.data_ptr(), _stream()),
                       'sslrec_infonce_bwd_scatter_f32')
            _infonce_record(ev, 'bwd', Bk, M, d, variant)
        # BPR rows of the clean view: added into the common table (SimGCL), or into a table of their own that the staging launch zeroes (SGL)
        wsb, kept = _bpr_bwd_ws(dev, B, d)
        fused_zero = (not simgcl) and kept and N * d % 4 == 0
        G3 = G2 if simgcl else (torch.empty_like(e0) if fused_zero else torch.zeros_like(e0))
        p3, q3 = v3.data_ptr(), G3.data_ptr()
        p3i, q3i = p3 + n_user * d * 4, q3 + n_user * d * 4
        args = (p3, ancs.data_ptr(), p3i, poss.data_ptr(), p3i, negs.data_ptr(), B, d, 0, float(B), c_one, q3, q3i, q3i, wsb.data_ptr())
        if kept:
            rc = lib.sslrec_bpr_bwd_kept_f32(*args, q3 if fused_zero else None, G3.numel() if fused_zero else 0, _stream())
        else:
            rc = lib.sslrec_bpr_bwd_f32(*args, _stream())
        _check_kept(rc, 'sslrec_bpr_bwd_f32', dev, B, d)
        reg_axpy = (e0, 2.0 * reg_weight, g)
        if simgcl:
            grad = _backward_chain(adjs[0], G2, L, d, axpy=reg_axpy)
        else:
            # (both sparse tables hold the rows ancs / n_user + poss / n_user + negs: SGL's item anchors are [poss; negs], sgl.py:58-59)
            rb3 = RowBits.from_indices(N, ancs, 0, poss, n_user, negs, n_user) if SPARSE_GRAD else None
            rb1 = rb3 if int(items_cl.numel()) == 2 * B else None
            o1 = _backward_chain(adjs[0], G1, L, d, sparse=rb1)
            o2 = _backward_chain(adjs[1], G2, L, d)
            graph3 = adjs[2].graph if isinstance(adjs[2], (DroppedView, RevaluedView)) else adjs[2]
            lay3 = graph3.bwd.swept(d) if graph3.bwd is not None else None
            fold = L >= 1 and lay3 is not None and bool(lib.sslrec_swept_deferred_sum_ok(C.byref(lay3.c_struct())))
            grad = _backward_chain(adjs[2], G3, L, d, axpy=reg_axpy, sparse=rb3, sum_in_last=[o1, o2] if fold else None)
            if not fold:
                _lib.check(lib.sslrec_add_tables_f32(grad.data_ptr(), o1.data_ptr(), o2.data_ptr(), grad.data_ptr(), grad.numel(), _stream()),
                           'sslrec_add_tables_f32')
        return grad[:n_user], grad[n_user:], None


def contrastive_step_ok(adjs, d, n_rows, batch_sizes):
    """can the SimGCL / SGL step run as one node?  (a kernel width for the InfoNCE, batches inside the scatter table, propagation on the
    column-swept or streamed kernels of a PropGraph with both directions)"""
    if not ONE_NODE_STEP or d not in INFONCE_DIMS or d not in SPMM_DIMS or 2 * max(batch_sizes) > 16384 or 3 * min(batch_sizes) > 16384:
        return False
    for a in adjs:
        g = a.graph if isinstance(a, (DroppedView, RevaluedView)) else a
        if not isinstance(g, PropGraph) or g.bwd is None or isinstance(a, RevaluedView):
            return False
    return True


def contrastive_step(user_embeds, item_embeds, spec):
    """(loss, bpr_loss, cl_loss, reg_loss) of a SimGCL / SGL step -- see _ContrastiveStepFn.  spec: kind 'simgcl' (adj, noises = two lists of L
    noise tables / tokens, eps) or 'sgl' (adjs = the three views' adjacencies); layer_num, ancs, poss, negs, items_cl (item anchors of the
    contrastive term: poss for SimGCL, [poss; negs] for SGL), temp, cl_weight, reg_weight, precision"""
    return _ContrastiveStepFn.apply(user_embeds, item_embeds, spec)


class _InfoNceShardedFn(torch.autograd.Function):
    """InfoNCE whose `all` rows are this rank's shard; `reduce(t)` sums a small tensor over the ranks in
    place (B floats forward, B*d floats backward).  Loss and dE1/dE2 come out identical on every rank."""

    @staticmethod
    def forward(ctx, e1, e2, all_local, temp, variant, reduce, will_differentiate=True):
        _need_gpu(e1, e2, all_local)
        e1, e2, all_local = _f32c(e1), _f32c(e2), _f32c(all_local)
        B, d = e1.shape
        M = all_local.shape[0]
        if d not in INFONCE_DIMS:
            raise ValueError('embedding size %d not supported by the HIP InfoNCE (supported: %s)' % (d, INFONCE_DIMS))
        if M == 0:
            raise ValueError('a rank holds no rows of the sharded table')
        lib = _lib.load()
        dev = e1.device
        if INFONCE_FWD_W and will_differentiate and any(ctx.needs_input_grad[:3]):
            variant |= INFONCE_FWD_W_BIT
        ws = torch.empty(lib.sslrec_infonce_ws_bytes(B, M, d) // 4, dtype=torch.float32, device=dev)
        z = torch.empty(B, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ev = _infonce_event()
        _lib.check(lib.sslrec_infonce_shard_rowsum_f32(e1.data_ptr(), 0, e2.data_ptr(), 0, B, all_local.data_ptr(), M, d,
                                                       float(temp), variant, ws.data_ptr(), z.data_ptr(), _stream()),
                   'sslrec_infonce_shard_rowsum_f32')
        reduce(z)
        _lib.check(lib.sslrec_infonce_shard_loss_f32(B, M, d, variant, ws.data_ptr(), z.data_ptr(), out.data_ptr(),
                                                     _stream()), 'sslrec_infonce_shard_loss_f32')
        _infonce_record(ev, 'fwd', B, M, d, variant)      # (the all-reduce of the B row sums between the two stages included)
        ctx.save_for_backward(ws)
        ctx.meta = (B, M, d, float(temp), variant, reduce)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (ws,) = ctx.saved_tensors
        B, M, d, temp, variant, reduce = ctx.meta
        g = g.reshape(1).to(torch.float32).contiguous()
        dev = ws.device
        w = torch.empty((B, d), dtype=torch.float32, device=dev)
        dall = torch.empty((M, d), dtype=torch.float32, device=dev)
        de1 = torch.empty((B, d), dtype=torch.float32, device=dev)
        de2 = torch.empty((B, d), dtype=torch.float32, device=dev)
        lib = _lib.load()
        ev = _infonce_event()
        _lib.check(lib.sslrec_infonce_shard_bwd_f32(B, M, d, temp, variant, ws.data_ptr(), g.data_ptr(), w.data_ptr(),
                                                    dall.data_ptr(), _stream()), 'sslrec_infonce_shard_bwd_f32')
        reduce(w)
        _lib.check(lib.sslrec_infonce_shard_finish_bwd_f32(B, M, d, temp, variant, ws.data_ptr(), g.data_ptr(),
                                                           w.data_ptr(), de1.data_ptr(), de2.data_ptr(), _stream()),
                   'sslrec_infonce_shard_finish_bwd_f32')
        _infonce_record(ev, 'bwd', B, M, d, variant)
        return de1, de2, dall, None, None, None, None


def infonce_loss_sharded(embeds1, embeds2, all_local, temp=1.0, variant=0, reduce=None, precision=None):
    """cal_infonce_loss(embeds1, embeds2, all, temp) (loss_utils.py:30-39) with `all` row-sharded:
    `all_local` = this rank's rows, embeds1/embeds2 = the same [B,d] rows on every rank, `reduce` =
    in-place sum over ranks (default: dist.all_reduce).  Returns the full SUM on every rank; the
    gradient w.r.t. all_local is complete for the local rows, w.r.t. embeds1/embeds2 the full one."""
    if reduce is None:
        import torch.distributed as dist
        reduce = dist.all_reduce
    dp = _padded_dim(embeds1.shape[1], INFONCE_DIMS)
    return _InfoNceShardedFn.apply(_pad_cols(embeds1, dp), _pad_cols(embeds2, dp), _pad_cols(all_local, dp), float(temp),
                                   _variant_code(variant, precision), reduce, torch.is_grad_enabled())


# ----------------------------------------------------------------------------------------------
# rank-q products of LightGCL's SVD view
# ----------------------------------------------------------------------------------------------
def _rankq_reduce(m, transposed, x):
    """sum_n M(.,n) x[n,:] -> [q,d];  m is [q,N] (transposed=False) or [N,q] (transposed=True), row-major"""
    q, n = (m.shape[1], m.shape[0]) if transposed else m.shape
    d = x.shape[1]
    lib = _lib.load()
    ws = torch.empty(lib.sslrec_rankq_ws_bytes(q, d) // 4, dtype=torch.float32, device=x.device)
    out = torch.empty((q, d), dtype=torch.float32, device=x.device)
    sq, sn = (1, q) if transposed else (n, 1)
    _lib.check(lib.sslrec_rankq_reduce_f32(m.data_ptr(), sq, sn, x.data_ptr(), n, d, q, ws.data_ptr(), out.data_ptr(), _stream()),
               'sslrec_rankq_reduce_f32')
    return out


def _rankq_expand(m, transposed, s_):
    """y[n,:] = sum_k M(k,n) s[k,:] -> [N,d];  m is [N,q] (transposed=True) or [q,N] (transposed=False), row-major"""
    q, n = (m.shape[1], m.shape[0]) if transposed else m.shape
    d = s_.shape[1]
    y = torch.empty((n, d), dtype=torch.float32, device=s_.device)
    sq, sn = (1, q) if transposed else (n, 1)
    _lib.check(_lib.load().sslrec_rankq_expand_f32(m.data_ptr(), sq, sn, s_.data_ptr(), n, d, q, y.data_ptr(), _stream()),
               'sslrec_rankq_expand_f32')
    return y


def rankq_reduce(m, transposed, x):
    """public form of the rank-q reduction (sharded LightGCL: partial products of a row shard, sslrec_amd/shard.py)"""
    _need_gpu(x, m)
    return _rankq_reduce(_f32c(m), transposed, _f32c(x))


def rankq_expand(m, transposed, s_):
    _need_gpu(s_, m)
    return _rankq_expand(_f32c(m), transposed, _f32c(s_))


class _LowRankFn(torch.autograd.Function):
    """left[N_out,q] @ (right[q,N_in] @ x[N_in,d]) with constant factors (LightGCL's `u_mul_s @ (vt @ E)`,
    lightgcl.py:83-84): two streaming kernels forward, the same two backward (dx = right^T (left^T dy))."""

    @staticmethod
    def forward(ctx, x, left, right):
        _need_gpu(x, left, right)
        x, left, right = _f32c(x), _f32c(left), _f32c(right)
        if left.shape[1] != right.shape[0] or right.shape[1] != x.shape[0] or left.shape[1] > 16:
            raise ValueError('low-rank factors %s, %s do not fit the operand %s (rank <= 16)' % (tuple(left.shape), tuple(right.shape), tuple(x.shape)))
        ctx.save_for_backward(left, right)
        return _rankq_expand(left, True, _rankq_reduce(right, False, x))

    @staticmethod
    def backward(ctx, gy):
        left, right = ctx.saved_tensors
        return _rankq_expand(right, False, _rankq_reduce(left, True, _f32c(gy))), None, None


def lowrank_apply(left, right, x):
    """left @ (right @ x) for rank-q factors (q <= 16), differentiable w.r.t. x"""
    return _LowRankFn.apply(x, left, right)


# ----------------------------------------------------------------------------------------------
# L2 regularizer term
# ----------------------------------------------------------------------------------------------
class _SumSqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        _need_gpu(x)
        x = _f32c(x)
        lib = _lib.load()
        ws = _ticket_ws(x.device, lib.sslrec_sumsq_ws_bytes(), 'sumsq')
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        _lib.check(lib.sslrec_sumsq_fwd_f32(x.data_ptr(), x.numel(), float(weight), ws.data_ptr(), out.data_ptr(), _stream()),
                   'sslrec_sumsq_fwd_f32')
        ctx.save_for_backward(x)
        ctx.weight = float(weight)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.reshape(1).to(torch.float32).contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.load().sslrec_sumsq_bwd_f32(x.data_ptr(), x.numel(), ctx.weight, g.data_ptr(), dx.data_ptr(), _stream()),
                   'sslrec_sumsq_bwd_f32')
        return dx, None


def sum_squares(x, weight=1.0):
    """weight * sum of squares of a parameter tensor (= reg_weight * W.norm(2).square() of reg_params,
    loss_utils.py:20-24, lightgcn.py:53) as one fused reduction, with the gradient 2*g*weight*W as one pass"""
    return _SumSqFn.apply(x, float(weight))


# ----------------------------------------------------------------------------------------------
# all-rank evaluation and negative sampling on the device
# ----------------------------------------------------------------------------------------------
def full_predict(user_table, item_table, users, train_mask=None):
    """[B, I] = (user_table[users] @ item_table.T) * (1 - train_mask) - 1e8 * train_mask: `full_predict` + `_mask_predict` of the
    reference (lightgcn.py:58-66, base_model.py:35-36) in one fused pass -- score tiles on the matrix cores, the mask read once, the
    result written once.  train_mask: [B, I] int64 (the reference's), float32, uint8 or bool, or None."""
    _need_gpu(user_table, item_table)
    ue, ie = _f32c(user_table), _f32c(item_table)
    d = ue.shape[1]
    if d > INFONCE_DIMS[-1]:
        raise ValueError('embedding size %d not supported by the HIP scoring kernel (up to %d)' % (d, INFONCE_DIMS[-1]))
    if d not in INFONCE_DIMS:
        dp = _padded_dim(d, INFONCE_DIMS)
        ue, ie, d = _f32c(_pad_cols(ue, dp)), _f32c(_pad_cols(ie, dp)), dp
    users = _idx(users)
    n_users = int(users.numel()) if users is not None else ue.shape[0]
    n_items = ie.shape[0]
    elem = 0
    if train_mask is not None:
        _need_gpu(train_mask)
        if tuple(train_mask.shape) != (n_users, n_items):
            raise ValueError('train mask of shape %s for %d users x %d items' % (tuple(train_mask.shape), n_users, n_items))
        if train_mask.dtype == torch.bool:
            train_mask = train_mask.view(torch.uint8)
        if train_mask.dtype not in (torch.int64, torch.float32, torch.uint8):
            train_mask = train_mask.to(torch.float32)
        train_mask = train_mask.contiguous()
        elem = train_mask.element_size()
    out = torch.empty((n_users, n_items), dtype=torch.float32, device=ue.device)
    rc = _lib.load().sslrec_full_predict_f32(ue.data_ptr(), _ptr(users), n_users, ie.data_ptr(), n_items, d, _ptr(train_mask), elem,
                                             out.data_ptr(), _stream())
    _lib.check(rc, 'sslrec_full_predict_f32')
    return out


def eval_topk(user_table, item_table, users, k, trn_csr=None, return_scores=False):
    """indices [B, k] (int64) of the k best-scoring items each user has NOT interacted with in training: the fused
    form of full_predict + _mask_predict + t.topk (lightgcn.py:58-66, base_model.py:35-36, metrics.py:99-103).
    `trn_csr` = (rowptr int64 [U+1], col int64 [nnz], sorted inside a row) on the device, or None."""
    _need_gpu(user_table, item_table)
    ue, ie = _f32c(user_table), _f32c(item_table)
    d = ue.shape[1]
    if d > INFONCE_DIMS[-1]:
        raise ValueError('embedding size %d not supported by the HIP evaluation kernel (up to %d)' % (d, INFONCE_DIMS[-1]))
    if int(k) > EVAL_KMAX:
        raise ValueError('k = %d exceeds the HIP evaluation kernel\'s per-user buffers (EVAL_KMAX = %d)' % (int(k), EVAL_KMAX))
    if d not in INFONCE_DIMS:       # other sizes (16 is in the reference's tuning grid): zero columns change no score
        dp = _padded_dim(d, INFONCE_DIMS)
        ue, ie, d = _f32c(_pad_cols(ue, dp)), _f32c(_pad_cols(ie, dp)), dp
    users = _idx(users)
    n_users = int(users.numel()) if users is not None else ue.shape[0]
    n_items = ie.shape[0]
    lib = _lib.load()
    ws = torch.empty(max(lib.sslrec_eval_topk_ws_bytes(n_users, n_items, int(k)), 8) // 4, dtype=torch.float32, device=ue.device)
    out = torch.empty((n_users, int(k)), dtype=torch.int64, device=ue.device)
    val = torch.empty((n_users, int(k)), dtype=torch.float32, device=ue.device) if return_scores else None
    rowptr, col = (None, None) if trn_csr is None else trn_csr
    rc = lib.sslrec_eval_topk_f32(ue.data_ptr(), _ptr(users), n_users, ie.data_ptr(), n_items, d, _ptr(rowptr), _ptr(col), int(k),
                                  ws.data_ptr(), out.data_ptr(), _ptr(val), _stream())
    _lib.check(rc, 'sslrec_eval_topk_f32')
    return (out, val) if return_scores else out


def sample_negs(users, trn_csr, n_item, philox_state, stream_id=None):
    """one negative item per interaction, uniform over the items the user has not interacted with (the distribution of
    datasets_general_cf.py:13-20), drawn on the device from the Philox stream of `philox_state` (sslrec_amd.rng)"""
    users = _idx(users)
    _need_gpu(users)
    rowptr, col = trn_csr
    out = torch.empty_like(users)
    sid = philox_state.next_stream() if stream_id is None else int(stream_id)
    rc = _lib.load().sslrec_sample_negs(users.data_ptr(), users.numel(), rowptr.data_ptr(), col.data_ptr(), int(n_item),
                                        philox_state.state.data_ptr(), sid, out.data_ptr(), _stream())
    _lib.check(rc, 'sslrec_sample_negs')
    return out
