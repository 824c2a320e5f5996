"""Row-sharded propagation over the GPUs of one node (one process per GPU, RCCL over xGMI
through torch.distributed; backend "nccl" IS RCCL on ROCm).

The reference has no multi-GPU code at all; this is the §8(e) design:

  * the N = U+I rows of the embedding table are dealt CYCLICALLY to the P ranks
    (row r lives on rank r % P at local index r // P), which balances the power-law
    degrees without any statistics; every rank keeps the CSR of its rows of A (and of
    A^T for the backward pass) with GLOBAL
    columns re-labelled into the all-gathered layout [rank][local index];
  * per layer ONE collective: all-gather of the previous layer's local rows (N/P * d * 4
    bytes per rank; cfg 2 at P=8: 4.6 MB shards, cfg 5: 1.28 GB shards = one shard per
    xGMI link), then the local HIP SpMM.  No floating-point reduction crosses a GPU, so
    the P-way result is BIT-IDENTICAL to the single-GPU result (same per-row summation
    order) -- the parity-friendly alternative to the all-reduce formulation;
  * backward mirrors it: all-gather of the local gradient rows + local SpMM with the
    A^T shard (g_{l-1} = G + A^T g_l).

The dual formulation the north star words as "all-reduce on the layer-wise propagated
embeddings" is provided as `mode='reduce_scatter'`: A is sharded by COLUMNS, every rank
multiplies its column slab with its own rows of X (no gather needed) into a full-height partial
result, and one reduce-scatter sums the partials and leaves each rank its rows.  Same bytes on
the wire, but a P-way fp32 sum in RCCL's order: equal to the single-GPU result to ~1e-7, not
bitwise, and each rank writes N*d partials instead of N/P*d -- so all-gather is the default.

`spmm_fn` is injectable so the partition / collective logic is testable on CPU with the
gloo backend (tests/test_shard_gloo.py feeds the oracle there); the default is the HIP op.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .graph import PropGraph, SEG_MAX


def local_rows(n, world, rank):
    """global ids of the rows rank `rank` owns (cyclic deal)"""
    return np.arange(rank, n, world, dtype=np.int64)


def rows_per_rank(n, world):
    return (n + world - 1) // world


def entry_key(rows, cols):
    """31-bit id of the matrix entry (row, col) -- a multiplicative mix of the pair, computable by whoever holds the entry"""
    x = np.asarray(rows, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.asarray(cols, dtype=np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    return (x >> np.uint64(33)).astype(np.int64)


def gathered_position(ids, n, world):
    """position of global row ids inside the all-gathered [rank][local] layout"""
    ids = np.asarray(ids, dtype=np.int64)
    return (ids % world) * rows_per_rank(n, world) + ids // world


class ShardedGraph:
    """This rank's slice of a square adjacency: the entries of ITS rows of A and of ITS rows of A^T.

    `ShardedGraph(rows, cols, vals, n, ...)` cuts them out of a global COO (tests, small graphs);
    `ShardedGraph.from_local_entries(...)` never sees anything global: a rank brings the pattern entries of its own rows of
    A and of A^T, the degrees are exchanged once (one all-gather of n/P counts) and the normalized values
    D^-1/2 A D^-1/2 (data_handler_general_cf.py:37-51) come out bit-identical to the data handler's."""

    def __init__(self, rows, cols, vals, n, world, rank, device, seg_max=SEG_MAX):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float32)
        f = np.nonzero(rows % world == rank)[0]            # entries of my rows of A
        b = np.nonzero(cols % world == rank)[0]            # entries of my rows of A^T
        # ids of the entries in the caller's COO: the element an EdgeDrop mask / Philox stream is defined over
        self._setup(n, world, rank, device, seg_max, (rows[f], cols[f], vals[f]), (rows[b], cols[b], vals[b]), f, b)

    @classmethod
    def from_local_entries(cls, fwd, bwd, n, world, rank, device, group=None, seg_max=SEG_MAX):
        """Shard-local construction of the normalized adjacency (no rank ever holds the global COO).  `fwd` = (rows, cols) of
        the binary pattern's entries whose ROW this rank owns (row % world == rank), `bwd` = those whose COLUMN it owns -- for the
        symmetric bipartite adjacency the same pairs mirrored.  Values: d^-1/2[row] * d^-1/2[col] with d = row count + 1e-10 in
        float64, cast to float32 -- the arithmetic of DataHandlerGeneralCF._normalize_adj.  EdgeDrop ids: a rank cannot know the
        position of its entries in the reference's global (column, row) order, so an entry's id is a 31-bit mix of (row, col):
        the same for the entry in an A shard and in an A^T shard, on every rank (perf-mode masks only; two entries in 2^31 may
        share a mask bit)."""
        (fr, fc), (br, bc) = [tuple(np.asarray(x, dtype=np.int64) for x in pair) for pair in (fwd, bwd)]
        if (fr.size and np.any(fr % world != rank)) or (bc.size and np.any(bc % world != rank)):
            raise ValueError('from_local_entries wants the entries of this rank\'s own rows (fwd) and columns (bwd)')
        n_per = rows_per_rank(n, world)
        deg_loc = np.bincount(fr // world, minlength=n_per).astype(np.float64)
        deg = _all_gather_host(deg_loc, world, group) + 1e-10                     # [rank][local] layout
        with np.errstate(divide='ignore'):
            dis = np.power(deg, -0.5)
        dis[np.isinf(dis)] = 0.0
        at = lambda ids: dis[gathered_position(ids, n, world)]
        vf = (at(fr) * at(fc)).astype(np.float32)
        vb = (at(br) * at(bc)).astype(np.float32)
        self = object.__new__(cls)
        self._setup(n, world, rank, device, seg_max, (fr, fc, vf), (br, bc, vb), entry_key(fr, fc), entry_key(br, bc))
        return self

    def _setup(self, n, world, rank, device, seg_max, fwd, bwd, ids_f, ids_b):
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        self.n_per = rows_per_rank(n, world)
        self.n_local = int(local_rows(n, world, rank).size)
        self.device = torch.device(device)
        n_gathered = self.n_per * world
        self.coo_ids_fwd, self.coo_ids_bwd = np.asarray(ids_f, dtype=np.int64), np.asarray(ids_b, dtype=np.int64)
        (fr, fc, fv), (br, bc, bv) = fwd, bwd
        # PropGraph(fwd = A_shard [n_per x n_gathered]); its own .bwd (transpose of the shard) is unused:
        # the backward pass needs ROWS of A^T, i.e. a second forward-type plan.
        self.a = PropGraph._single(fr // world, fc, fv, (self.n_per, n_gathered), device, seg_max,
                                   col_relabel=lambda c: gathered_position(c, n, world))
        self.at = PropGraph._single(bc // world, br, bv, (self.n_per, n_gathered), device, seg_max,
                                    col_relabel=lambda c: gathered_position(c, n, world))
        self.nnz_local = int(fr.size)
        self._col_sharded = None
        self._blocks = None
        self._loc = (fwd, bwd, seg_max)

    def source_blocks(self):
        """(A blocks, A^T blocks): block q = my rows x the columns owned by rank q (local column ids), the operands of the
        PIPELINED exchange: the shard of rank q is multiplied as soon as it has arrived, while the later shards are still
        on the wire (SURVEY.md §8e "overlap"); built on first use from this rank's own entries"""
        if self._blocks is None:
            (fr, fc, fv), (br, bc, bv), seg_max = self._loc
            world = self.world
            out = []
            for r_, c_, v_ in ((fr, fc, fv), (bc, br, bv)):            # (my row, its column, value) of A / of A^T
                blocks = []
                for q in range(world):
                    sel = np.nonzero(c_ % world == q)[0]
                    blocks.append(PropGraph._single(r_[sel] // world, c_[sel] // world, v_[sel], (self.n_per, self.n_per),
                                                    self.device, seg_max))
                out.append(blocks)
            self._blocks = tuple(out)
        return self._blocks

    def dropped(self, keep_rate, philox_state, stream, scale=1.0):
        """this rank's shards of the EDGE-DROPPED adjacency (EdgeDrop, aug_utils.py:18-31, in perf mode): the mask bit of
        COO entry k is a pure function of (seed, step, stream, k) with k the GLOBAL entry id, so every rank -- and the A /
        A^T shards of one rank -- drop the same edges without exchanging anything.  Returns an object usable wherever
        the ShardedGraph itself is (sharded_propagate_sum)."""
        from .graph import DroppedView
        dev = self.device
        if not hasattr(self, '_ids_dev'):
            self._ids_dev = (torch.from_numpy(self.coo_ids_fwd.astype(np.int64)).to(dev),
                             torch.from_numpy(self.coo_ids_bwd.astype(np.int64)).to(dev))
        view = _ShardView(self)
        view.a = DroppedView(self.a, None, scale, philox=(philox_state, stream, keep_rate), entry_ids=self._ids_dev[0])
        view.at = DroppedView(self.at, None, scale, philox=(philox_state, stream, keep_rate), entry_ids=self._ids_dev[1])
        return view

    def col_sharded(self):
        """(A[:, my cols], A^T[:, my cols]) with rows re-labelled into the [rank][local] layout -- the
        operands of the reduce-scatter formulation; built on first use (the entries whose COLUMN this rank owns are the
        entries of its rows of the transposed matrix: both lists are local)."""
        if self._col_sharded is None:
            (fr, fc, fv), (br, bc, bv), seg_max = self._loc
            n, world = self.n, self.world
            n_gathered = self.n_per * world
            a_c = PropGraph._single(gathered_position(br, n, world), bc // world, bv,
                                    (n_gathered, self.n_per), self.device, seg_max)
            at_c = PropGraph._single(gathered_position(fc, n, world), fr // world, fv,
                                     (n_gathered, self.n_per), self.device, seg_max)
            self._col_sharded = (a_c, at_c)
        return self._col_sharded

    def to_local(self, full):
        """rows of a full [N, d] host/device tensor owned by this rank, padded to n_per rows"""
        ids = torch.from_numpy(local_rows(self.n, self.world, self.rank)).to(full.device)
        out = torch.zeros((self.n_per, full.shape[1]), dtype=full.dtype, device=full.device)
        out[:ids.numel()] = full[ids]
        return out


def solo(world):
    """True when a collective over `world` ranks is the identity and may be skipped.  SSLREC_FORCE_COLLECTIVES=1 switches every
    such short cut off, so that a ONE-rank process group still issues each collective of the N > 1 path -- how the RCCL calls
    (backend "nccl") are exercised on a box with a single GPU (tests/test_gpu_parity.py, bench.py under WORLD_SIZE=1)."""
    return world == 1 and os.environ.get('SSLREC_FORCE_COLLECTIVES', '0') != '1'


def _host_staged(group, t):
    """gloo moves host memory: device tensors are staged through the host (the correctness path for running several
    ranks on ONE GPU, tests/test_gpu_parity.py; RCCL -- backend "nccl" -- takes device tensors directly)"""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


class _ShardView:
    """a ShardedGraph whose two shard matrices are replaced by views (edge-dropped): same partition, same collectives"""

    def __init__(self, sg):
        self.world, self.rank, self.n, self.n_per, self.n_local, self.device = sg.world, sg.rank, sg.n, sg.n_per, sg.n_local, sg.device


def all_gather_rows(x_local, world, group=None, async_op=False):
    """[n_per, d] per rank -> [world * n_per, d] in [rank][local] order (one collective).  With async_op the
    collective is only enqueued: returns (out, wait) and `wait()` must be called before `out` is read."""
    if solo(world):
        return (x_local, lambda: None) if async_op else x_local
    x_local = x_local.contiguous()
    if _host_staged(group, x_local):
        host = x_local.cpu()
        out_h = torch.empty((world * host.shape[0], host.shape[1]), dtype=host.dtype)
        dist.all_gather_into_tensor(out_h, host, group=group)
        out = out_h.to(x_local.device)
        return (out, lambda: None) if async_op else out
    out = torch.empty((world * x_local.shape[0], x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
    if async_op:
        work = dist.all_gather_into_tensor(out, x_local, group=group, async_op=True)
        return out, work.wait
    dist.all_gather_into_tensor(out, x_local, group=group)
    return out


def all_reduce_sum(t, group=None):
    """in-place sum over the ranks (host-staged under gloo for device tensors)"""
    if _host_staged(group, t):
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


def reduce_scatter_rows(y_full, world, group=None):
    """[world * n_per, d] partial results per rank -> this rank's [n_per, d] rows of their sum"""
    if solo(world):
        return y_full
    y_full = y_full.contiguous()
    if dist.get_backend(group) == 'gloo':          # gloo has no reduce-scatter: all-reduce + slice (tests only)
        host = y_full.cpu() if y_full.is_cuda else y_full.clone()
        dist.all_reduce(host, group=group)
        n_per = y_full.shape[0] // world
        r = dist.get_rank(group)
        return host[r * n_per:(r + 1) * n_per].to(y_full.device)
    out = torch.empty((y_full.shape[0] // world, y_full.shape[1]), dtype=y_full.dtype, device=y_full.device)
    dist.reduce_scatter_tensor(out, y_full, op=dist.ReduceOp.SUM, group=group)
    return out


def _default_spmm(plan_graph, x_gathered, acc_in, acc_out, want_y, noise=None, eps=0.0):
    return ops.spmm_raw(plan_graph, x_gathered, 'fwd', noise=noise, eps=eps, acc_in=acc_in, acc_out=acc_out,
                        want_y=want_y)


class _ShardedPropagateSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e0_local, sg, layer_num, spmm_fn, group, noises=None, eps=0.0):
        ctx.sg, ctx.layer_num, ctx.spmm_fn, ctx.group = sg, layer_num, spmm_fn, group
        e0_local = e0_local.contiguous()
        if layer_num == 0:
            return e0_local.clone()
        total = torch.empty_like(e0_local)
        x = e0_local
        for l in range(layer_num):
            xg = all_gather_rows(x, sg.world, group)
            last = (l == layer_num - 1)
            if noises is None:
                x = spmm_fn(sg.a, xg, e0_local if l == 0 else total, total, not last)
            else:           # EmbedPerturb fused into the epilogue (simgcl.py:25-27); its gradient is the identity
                x = spmm_fn(sg.a, xg, e0_local if l == 0 else total, total, not last, noise=noises[l], eps=eps)
        return total

    @staticmethod
    def backward(ctx, g_total):
        sg = ctx.sg
        g_total = g_total.contiguous()
        g = g_total
        for _ in range(ctx.layer_num):
            gg = all_gather_rows(g, sg.world, ctx.group)
            nxt = torch.empty_like(g_total)
            ctx.spmm_fn(sg.at, gg, g_total, nxt, False)
            g = nxt
        return g, None, None, None, None, None, None


class _ShardedPropagateSumViewsFn(torch.autograd.Function):
    """K views of the layer-summed propagation of the SAME local rows over the SAME adjacency, differing only in their per-layer
    perturbation noise (None = clean): SimGCL's three forwards (simgcl.py:29-31) on row-sharded tables, the counterpart of
    ops._PropagateSumViewsFn.  E0 is all-gathered ONCE for the K first-layer products, and -- the perturbation's derivative being the
    identity, every view's backward pass is the same linear map -- the backward chain runs once on the SUM of the views' upstream
    gradients: L exchanges of gradient rows per step instead of K L (K = 3, L = 2: 6 -> 2), K L - (K - 1) forward exchanges instead of K L."""

    @staticmethod
    def forward(ctx, e0_local, sg, layer_num, spmm_fn, group, noises_views, eps):
        ctx.sg, ctx.layer_num, ctx.spmm_fn, ctx.group = sg, layer_num, spmm_fn, group
        e0_local = e0_local.contiguous()
        if layer_num == 0:
            return tuple(e0_local.clone() for _ in noises_views)
        xg0 = all_gather_rows(e0_local, sg.world, group)
        totals = []
        for nz in noises_views:
            total = torch.empty_like(e0_local)
            x = None
            for l in range(layer_num):
                xg = xg0 if l == 0 else all_gather_rows(x, sg.world, group)
                last = (l == layer_num - 1)
                if nz is None:
                    x = spmm_fn(sg.a, xg, e0_local if l == 0 else total, total, not last)
                else:
                    x = spmm_fn(sg.a, xg, e0_local if l == 0 else total, total, not last, noise=nz[l], eps=eps)
            totals.append(total)
        return tuple(totals)

    @staticmethod
    def backward(ctx, *g_totals):
        sg = ctx.sg
        grads = [g.contiguous() for g in g_totals if g is not None]
        if not grads:
            return (None,) * 7
        G = grads[0]
        for i, g in enumerate(grads[1:]):
            G = torch.add(G, g) if i == 0 else G.add_(g)
        g = G
        for _ in range(ctx.layer_num):
            gg = all_gather_rows(g, sg.world, ctx.group)
            nxt = torch.empty_like(G)
            ctx.spmm_fn(sg.at, gg, G, nxt, False)
            g = nxt
        return g, None, None, None, None, None, None


def sharded_propagate_sum_views(sg, e0_local, layer_num, noises_views, eps=0.0, spmm_fn=None, group=None):
    """[sharded_propagate_sum(sg, e0_local, L, noises=nz, eps=eps) for nz in noises_views] (all_gather formulation) with the first
    exchange and the whole backward chain shared between the views"""
    return list(_ShardedPropagateSumViewsFn.apply(e0_local, sg, int(layer_num), spmm_fn or _default_spmm, group,
                                                  [None if nz is None else list(nz) for nz in noises_views], float(eps)))


def shards_pipelined(x_local, world, rank, group=None):
    """the P row shards of x as they arrive: yields (q, shard of rank q), own shard first, the others in ring order
    after it; every shard travels as its own broadcast, enqueued up front, so the consumer works on shard q while the
    later ones are still in flight (under RCCL `wait()` only orders the streams)"""
    x_local = x_local.contiguous()
    if solo(world):
        yield 0, x_local
        return
    staged = _host_staged(group, x_local)
    send = x_local.cpu() if staged else x_local
    bufs, works = {}, {}
    for q in range(world):                       # every rank issues the broadcasts in the same order
        bufs[q] = send if q == rank else torch.empty_like(send)
        works[q] = dist.broadcast(bufs[q], src=q, group=group, async_op=True)
    yield rank, x_local
    for k in range(1, world):
        q = (rank + k) % world
        works[q].wait()
        yield q, (bufs[q].to(x_local.device) if staged else bufs[q])
    works[rank].wait()


def _pipelined_product(blocks, x_local, world, rank, spmm_fn, group):
    """sum_q A[my rows, cols of q] @ x_q with the exchange pipelined against the block products"""
    y = None
    for q, xq in shards_pipelined(x_local, world, rank, group):
        if y is None:
            y = spmm_fn(blocks[q], xq, None, None, True)
        else:
            spmm_fn(blocks[q], xq, y, y, False)
    return y


class _ShardedPropagateSumPipeFn(torch.autograd.Function):
    """the all-gather formulation with the exchange pipelined against per-source-rank block products; the row sums
    are formed source rank by source rank, so the result equals the single-GPU one to rounding (~1e-7), not bitwise"""

    @staticmethod
    def forward(ctx, e0_local, sg, layer_num, spmm_fn, group):
        ctx.sg, ctx.layer_num, ctx.spmm_fn, ctx.group = sg, layer_num, spmm_fn, group
        a_blocks, _ = sg.source_blocks()
        e0_local = e0_local.contiguous()
        total, x = e0_local.clone(), e0_local
        for _ in range(layer_num):
            x = _pipelined_product(a_blocks, x, sg.world, sg.rank, spmm_fn, group)
            total += x
        return total

    @staticmethod
    def backward(ctx, g_total):
        sg = ctx.sg
        _, at_blocks = sg.source_blocks()
        g_total = g_total.contiguous()
        g = g_total
        for _ in range(ctx.layer_num):
            g = g_total + _pipelined_product(at_blocks, g, sg.world, sg.rank, ctx.spmm_fn, ctx.group)
        return g, None, None, None, None


class _ShardedPropagateSumRsFn(torch.autograd.Function):
    """reduce-scatter formulation (column-sharded A): partial = A[:, mine] @ x_local; y = RS(partial)"""

    @staticmethod
    def forward(ctx, e0_local, sg, layer_num, spmm_fn, group):
        ctx.sg, ctx.layer_num, ctx.spmm_fn, ctx.group = sg, layer_num, spmm_fn, group
        a_c, _ = sg.col_sharded()
        e0_local = e0_local.contiguous()
        total = e0_local.clone()
        x = e0_local
        for _ in range(layer_num):
            partial = spmm_fn(a_c, x, None, None, True)
            x = reduce_scatter_rows(partial, sg.world, group)
            total += x
        return total

    @staticmethod
    def backward(ctx, g_total):
        sg = ctx.sg
        _, at_c = sg.col_sharded()
        g_total = g_total.contiguous()
        g = g_total
        for _ in range(ctx.layer_num):
            partial = ctx.spmm_fn(at_c, g, None, None, True)
            g = g_total + reduce_scatter_rows(partial, sg.world, ctx.group)
        return g, None, None, None, None


def sharded_propagate_sum(sg, e0_local, layer_num, spmm_fn=None, group=None, mode='all_gather', noises=None, eps=0.0):
    """Local rows of  E0 + sum_l A^l E0  for a row-sharded table (differentiable).
    mode: 'all_gather' (row-sharded A, bit-identical to one GPU), 'pipelined' (the same exchange as P broadcasts
    overlapped with per-source-rank block products) or 'reduce_scatter' (column-sharded A).
    noises: optional list of L local [n_per, d] uniform draws -> SimGCL's per-layer perturbation with
    magnitude eps, fused into the local SpMM's epilogue (all_gather mode)."""
    if mode in ('reduce_scatter', 'pipelined'):
        if noises is not None:
            raise ValueError('perturbed propagation is implemented for the all_gather formulation')
        fn = _ShardedPropagateSumRsFn if mode == 'reduce_scatter' else _ShardedPropagateSumPipeFn
        return fn.apply(e0_local, sg, int(layer_num), spmm_fn or _default_spmm, group)
    return _ShardedPropagateSumFn.apply(e0_local, sg, int(layer_num), spmm_fn or _default_spmm, group,
                                        None if noises is None else list(noises), float(eps))


# ----------------------------------------------------------------------------------------------
# losses on row-sharded tables: batch-parallel over the all-gathered final embeddings
# ----------------------------------------------------------------------------------------------
class _AllGatherRowsFn(torch.autograd.Function):
    """differentiable all-gather of row shards: backward = reduce-scatter (sum) of the gradient"""

    @staticmethod
    def forward(ctx, x_local, world, group):
        ctx.world, ctx.group = world, group
        return all_gather_rows(x_local, world, group)

    @staticmethod
    def backward(ctx, g_full):
        return reduce_scatter_rows(g_full.contiguous(), ctx.world, ctx.group), None, None


class _ExchangeRowsFn(torch.autograd.Function):
    """rows `ids` (global ids) of a row-sharded table, identical on every rank.  Forward: every rank fills
    the rows it owns into a zero [K,d] buffer, one all-reduce (exact: x + 0 + ... + 0).  Backward: the
    caller computes the same loss on every rank, so the incoming gradient is already the full one --
    each rank keeps the rows it owns (index_add into its shard), no collective."""

    @staticmethod
    def forward(ctx, s_local, ids, world, rank, group):
        loc = torch.div(ids, world, rounding_mode='floor')
        mine = (ids - loc * world) == rank
        buf = torch.where(mine[:, None], s_local.index_select(0, loc), torch.zeros((), dtype=s_local.dtype, device=s_local.device))
        if not solo(world):
            all_reduce_sum(buf, group)
        ctx.save_for_backward(loc, mine)
        ctx.n_rows = s_local.shape[0]
        return buf

    @staticmethod
    def backward(ctx, g):
        loc, mine = ctx.saved_tensors
        g = torch.where(mine[:, None], g, torch.zeros((), dtype=g.dtype, device=g.device))
        out = torch.zeros((ctx.n_rows, g.shape[1]), dtype=g.dtype, device=g.device)
        out.index_add_(0, loc, g)
        return out, None, None, None, None


class _ShardedPropagateRowsFn(torch.autograd.Function):
    """rows `ids` of the layer-summed propagated table (identical on every rank) in ONE node, so that the backward pass
    can skip a collective: every rank evaluates the same batch loss, hence holds the same gradient for the same rows,
    and builds the all-gathered gradient of the first backward layer locally instead of all-gathering 37 MB of mostly
    zero rows.  Collectives per step: L all-gathers forward, one small all-reduce, L-1 all-gathers backward."""

    @staticmethod
    def forward(ctx, e0_local, ids, sg, layer_num, spmm_fn, group):
        total = _ShardedPropagateSumFn.forward(ctx, e0_local, sg, layer_num, spmm_fn, group)
        loc = torch.div(ids, sg.world, rounding_mode='floor')
        mine = (ids - loc * sg.world) == sg.rank
        buf = torch.where(mine[:, None], total.index_select(0, loc), torch.zeros((), dtype=total.dtype, device=total.device))
        if not solo(sg.world):
            all_reduce_sum(buf, group)
        ctx.save_for_backward(ids, loc, mine)
        ctx.n_per = total.shape[0]
        return buf

    @staticmethod
    def backward(ctx, g_rows):
        ids, loc, mine = ctx.saved_tensors
        sg, L = ctx.sg, ctx.layer_num
        g_rows = g_rows.contiguous()
        d = g_rows.shape[1]
        zero = torch.zeros((), dtype=g_rows.dtype, device=g_rows.device)
        g_local = torch.zeros((ctx.n_per, d), dtype=g_rows.dtype, device=g_rows.device)
        g_local.index_add_(0, loc, torch.where(mine[:, None], g_rows, zero))            # gradient of this rank's rows of `total`
        if L == 0:
            return g_local, None, None, None, None, None
        # first backward layer: the gathered gradient is known everywhere -- no all-gather
        gathered = torch.zeros((ctx.n_per * sg.world, d), dtype=g_rows.dtype, device=g_rows.device)
        gathered.index_add_(0, (ids - loc * sg.world) * ctx.n_per + loc, g_rows)
        g = torch.empty_like(g_local)
        ctx.spmm_fn(sg.at, gathered, g_local, g, False)
        for _ in range(L - 1):
            gg = all_gather_rows(g, sg.world, ctx.group)
            nxt = torch.empty_like(g_local)
            ctx.spmm_fn(sg.at, gg, g_local, nxt, False)
            g = nxt
        return g, None, None, None, None, None


class ShardedGraphCF(torch.nn.Module):
    """LightGCN-family model whose stacked embedding table [users; items] is ROW-SHARDED over the
    ranks (parameter = this rank's rows, so optimizer state is sharded too).

    Training step = sharded propagation (one all-gather per layer, see above) -> the 3B batch rows are
    exchanged with ONE small all-reduce (`rows()`, B*3*d*4 bytes instead of the whole table) -> every rank
    evaluates the batch losses on the same rows (BPR: microseconds) -> the backward of the exchange is
    local.  InfoNCE keeps `all` sharded (`infonce()`): each rank streams its own rows, the B row sums and
    the B x d anchor gradients are all-reduced (ops.infonce_loss_sharded).
    Evaluation keeps the item table sharded too (`predict_topk`: per-rank fused top-k + a merge of P lists of k);
    `tables()` (full tables on every rank, two table-sized collectives) remains for callers that want them.

    Loss values: the batch terms are identical on all ranks; the regularizer is this rank's share
    (`last_parts['reg_local']`) -- all-reduce it for logging.
    """

    def __init__(self, sg, n_user, n_item, init_table, layer_num, spmm_fn=None, group=None, mode='all_gather'):
        super().__init__()
        self.sg, self.n_user, self.n_item, self.layer_num = sg, int(n_user), int(n_item), int(layer_num)
        self.spmm_fn, self.group, self.mode = spmm_fn, group, mode
        self.local_embeds = torch.nn.Parameter(sg.to_local(init_table).to(sg.device))
        pos = gathered_position(np.arange(sg.n), sg.n, sg.world)
        self.register_buffer('pos_users', torch.from_numpy(pos[:self.n_user]).to(sg.device), persistent=False)
        self.register_buffer('pos_items', torch.from_numpy(pos[self.n_user:]).to(sg.device), persistent=False)
        # users are a prefix of the local shard: global id g = rank + k*world < n_user  <=>  k < k_user
        self.k_user = max(0, -(-(self.n_user - sg.rank) // sg.world))
        self.last_parts = {}

    # ---- propagation -------------------------------------------------------------------------
    def propagate(self, noises=None, eps=0.0):
        """this rank's rows of the layer-summed propagated table (differentiable)"""
        return sharded_propagate_sum(self.sg, self.local_embeds, self.layer_num, self.spmm_fn, self.group, self.mode,
                                     noises=noises, eps=eps)

    def tables(self):
        """(user table [U,d], item table [I,d]) of the propagated + layer-summed embeddings, full
        and in global row order on every rank.  Differentiable under a BATCH-PARALLEL contract only: the backward is a
        reduce-scatter SUM of the ranks' gradients, which is right when every rank evaluates a DIFFERENT slice of the
        batch.  Code that evaluates the SAME loss on every rank (rows(), lightgcn_loss, simgcl_loss do) must not
        differentiate through tables() -- it would receive world_size times the gradient; use rows()."""
        s_all = _AllGatherRowsFn.apply(self.propagate(), self.sg.world, self.group)
        return s_all.index_select(0, self.pos_users), s_all.index_select(0, self.pos_items)

    # ---- evaluation without gathering the tables ---------------------------------------------------------------------
    def local_item_ids(self):
        """global item ids of this rank's item rows, in local order (ascending)"""
        j = np.arange(self.sg.n_local - self.k_user, dtype=np.int64)
        return self.sg.rank + (self.k_user + j) * self.sg.world - self.n_user

    def local_train_csr(self, trn_mat):
        """(rowptr, col) on the device of the train interactions restricted to this rank's items, columns = LOCAL item
        positions, sorted inside a row: the mask operand of predict_topk (built once per rank)"""
        sub = trn_mat.tocsr()[:, self.local_item_ids()].tocsr()
        sub.sort_indices()
        dev = self.sg.device
        return (torch.from_numpy(sub.indptr.astype(np.int64)).to(dev), torch.from_numpy(sub.indices.astype(np.int64)).to(dev))

    def predict_topk(self, users, k, local_trn=None, topk_fn=None):
        """All-rank evaluation of a batch of users with the item table kept SHARDED (the reference's full_predict +
        _mask_predict + t.topk, lightgcn.py:58-66, base_model.py:35-36, trainer/metrics.py:99-103): the batch's user rows
        are exchanged (one B x d all-reduce), every rank runs the fused top-k kernel over ITS items and the train
        interactions among them, and the P lists of k (score, global item id) pairs are all-gathered and merged --
        2 P B k numbers on the wire instead of the two tables.  Same result on every rank: ids [B, k] (descending score,
        ties to the smaller item id, -1 where a user has fewer than k unseen items) and scores."""
        sg = self.sg
        users = users.to(sg.device).long()
        B = int(users.numel())
        with torch.no_grad():
            s_local = self.propagate()
            ue = self.rows(s_local, users).contiguous()
            ie = self.local_items(s_local).contiguous()
            csr = None
            if local_trn is not None:          # the batch's rows of the local train CSR
                rowptr, col = local_trn
                lo, hi = rowptr[users], rowptr[users + 1]
                lens = hi - lo
                new_ptr = torch.zeros(B + 1, dtype=torch.int64, device=sg.device)
                new_ptr[1:] = lens.cumsum(0)
                within = torch.arange(int(new_ptr[-1].item()), device=sg.device) - new_ptr[:-1].repeat_interleave(lens)
                csr = (new_ptr, col[lo.repeat_interleave(lens) + within])
            if ie.shape[0] > 0:
                idx, val = (topk_fn or ops.eval_topk)(ue, ie, None, k, csr, return_scores=True)
            else:                                # a rank without item rows (tiny tables)
                idx = torch.full((B, k), -1, dtype=torch.int64, device=sg.device)
                val = torch.full((B, k), float('-inf'), device=sg.device)
            gid = torch.where(idx >= 0, sg.rank + (self.k_user + idx) * sg.world - self.n_user, idx)
            vals = all_gather_rows(val.float().contiguous(), sg.world, self.group).view(sg.world, B, k)
            gids = all_gather_rows(gid.view(B, k).double().contiguous(), sg.world, self.group).view(sg.world, B, k).long()
            vals = vals.permute(1, 0, 2).reshape(B, sg.world * k)
            gids = gids.permute(1, 0, 2).reshape(B, sg.world * k)
            key = torch.where(gids >= 0, gids, torch.full_like(gids, 2 ** 62))             # missing entries last
            by_id = key.argsort(dim=1, stable=True)
            vals, gids = vals.gather(1, by_id), gids.gather(1, by_id)
            by_val = (-vals).argsort(dim=1, stable=True)                                    # stable: ties keep the id order
            return gids.gather(1, by_val)[:, :k], vals.gather(1, by_val)[:, :k]

    def local_users(self, s_local):
        return s_local[:self.k_user]

    def local_items(self, s_local):
        return s_local[self.k_user:self.sg.n_local]

    def rows(self, s_local, ids):
        """rows of the stacked table for global stacked ids, same on every rank (one small all-reduce)"""
        return _ExchangeRowsFn.apply(s_local, ids, self.sg.world, self.sg.rank, self.group)

    def batch_rows(self, s_local, batch):
        """(anchor, positive, negative) rows [B,d] each of a (users, pos items, neg items) batch"""
        ancs, poss, negs = batch[:3]
        B = ancs.shape[0]
        buf = self.rows(s_local, torch.cat([ancs, poss + self.n_user, negs + self.n_user]))
        return buf[:B], buf[B:2 * B], buf[2 * B:]

    # ---- losses ------------------------------------------------------------------------------
    def reg_loss(self, reg_fn=None):
        """sum of squares of the LOCAL rows (padding rows are zero and stay zero)"""
        return (reg_fn or ops.sum_squares)(self.local_embeds)

    def infonce(self, e1, e2, all_local, temp, infonce_fn=None):
        """cal_infonce_loss(e1, e2, all, temp) with `all` = the concatenation of every rank's all_local"""
        if infonce_fn is not None:
            return infonce_fn(e1, e2, all_local, temp)
        grp = self.group
        red = (lambda t: t) if solo(self.sg.world) else (lambda t: all_reduce_sum(t, grp))
        return ops.infonce_loss_sharded(e1, e2, all_local, temp, 0, red)

    def lightgcn_loss(self, batch, reg_weight, bpr_fn=None, reg_fn=None):
        """LightGCN's loss (reference lightgcn.py:45-56): bpr/B (full, same on every rank) +
        reg_weight * (this rank's share of the regularizer)"""
        if self.mode == 'all_gather':          # propagation + row exchange as one node: one collective fewer in backward
            ancs, poss, negs = batch[:3]
            B = ancs.shape[0]
            ids = torch.cat([ancs, poss + self.n_user, negs + self.n_user])
            buf = _ShardedPropagateRowsFn.apply(self.local_embeds, ids, self.sg, self.layer_num,
                                                self.spmm_fn or _default_spmm, self.group)
            anc, pos, neg = buf[:B], buf[B:2 * B], buf[2 * B:]
        else:
            anc, pos, neg = self.batch_rows(self.propagate(), batch)
        bpr = (bpr_fn or ops.bpr_loss)(anc, pos, neg) / batch[0].shape[0]
        reg = self.reg_loss(reg_fn)
        self.last_parts = {'bpr_loss': bpr.detach(), 'reg_local': reg.detach()}
        return bpr + reg_weight * reg

    def sgl_loss(self, batch, keep_rate, philox_state, reg_weight, cl_weight, temp, bpr_fn=None, reg_fn=None, infonce_fn=None):
        """SGL-ED's loss (reference sgl.py:45-65) on the sharded table: two independently edge-dropped propagations (masks
        computed in the kernels from the GLOBAL entry ids, ShardedGraph.dropped) and the clean one, BPR on the clean view,
        three InfoNCE terms with `all` kept sharded.  Call philox_state.advance() once per step before."""
        ancs, poss, negs = batch[:3]
        B = ancs.shape[0]
        spmm = self.spmm_fn
        views = [self.sg.dropped(keep_rate, philox_state, philox_state.next_stream()) for _ in range(2)]
        v1, v2 = (sharded_propagate_sum(v, self.local_embeds, self.layer_num, spmm, self.group) for v in views)
        anc, pos, neg = self.batch_rows(self.propagate(), batch)
        bpr = (bpr_fn or ops.bpr_loss)(anc, pos, neg) / B
        ids = torch.cat([ancs, poss + self.n_user, negs + self.n_user])
        r1, r2 = self.rows(v1, ids), self.rows(v2, ids)
        cl = self.infonce(r1[:B], r2[:B], self.local_users(v2), temp, infonce_fn) + \
            self.infonce(r1[B:], r2[B:], self.local_items(v2), temp, infonce_fn)      # positives and negatives: one call, same `all` (sgl.py:58-59)
        cl = cl / B
        reg = self.reg_loss(reg_fn)
        self.last_parts = {'bpr_loss': bpr.detach(), 'cl_loss': cl.detach(), 'reg_local': reg.detach()}
        return bpr + reg_weight * reg + cl_weight * cl

    def simgcl_loss(self, batch, noises1, noises2, eps, reg_weight, cl_weight, temp, bpr_fn=None, reg_fn=None,
                    infonce_fn=None):
        """SimGCL's loss (reference simgcl.py:29-54) on the sharded table: one clean and two perturbed
        propagations (noises*: L local [n_per,d] uniform draws each), BPR on the clean view, InfoNCE
        between the perturbed views for the batch users against all users and the batch positives against
        all items."""
        ancs, poss = batch[0], batch[1]
        B = ancs.shape[0]
        if self.mode == 'all_gather':      # one exchange of E0 for the three first products, one backward chain for the three views
            v1, v2, v3 = sharded_propagate_sum_views(self.sg, self.local_embeds, self.layer_num, [noises1, noises2, None], eps,
                                                     self.spmm_fn, self.group)
        else:
            v1, v2, v3 = self.propagate(noises1, eps), self.propagate(noises2, eps), self.propagate()
        anc, pos, neg = self.batch_rows(v3, batch)
        bpr = (bpr_fn or ops.bpr_loss)(anc, pos, neg) / B
        ids = torch.cat([ancs, poss + self.n_user])
        r1, r2 = self.rows(v1, ids), self.rows(v2, ids)
        cl = self.infonce(r1[:B], r2[:B], self.local_users(v2), temp, infonce_fn) + \
            self.infonce(r1[B:], r2[B:], self.local_items(v2), temp, infonce_fn)
        cl = cl / B
        reg = self.reg_loss(reg_fn)
        self.last_parts = {'bpr_loss': bpr.detach(), 'cl_loss': cl.detach(), 'reg_local': reg.detach()}
        return bpr + reg_weight * reg + cl_weight * cl


# ----------------------------------------------------------------------------------------------
# LightGCL on row-sharded tables (BASELINE config 5: U x I biadjacency, embeddings row-sharded over the GPUs)
# ----------------------------------------------------------------------------------------------
class ShardedBipartite:
    """This rank's slices of a U x I adjacency A given as global COO (users, items, vals): the rows of A owned by the
    rank (users dealt cyclically, columns = items in the all-gathered [rank][local] layout) and the rows of A^T owned
    by it (items dealt cyclically, columns = gathered users).  Reference: the model-private adjacency of
    models/general_cf/lightgcl.py:16-22 and its two products per layer (:78-79); the reference has no multi-GPU path."""

    def __init__(self, users, items, vals, n_user, n_item, world, rank, device, seg_max=SEG_MAX):
        users = np.asarray(users, dtype=np.int64)
        items = np.asarray(items, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float32)
        self.n_user, self.n_item, self.world, self.rank = int(n_user), int(n_item), int(world), int(rank)
        self.u_per, self.i_per = rows_per_rank(n_user, world), rows_per_rank(n_item, world)
        self.u_local = int(local_rows(n_user, world, rank).size)
        self.i_local = int(local_rows(n_item, world, rank).size)
        self.device = torch.device(device)
        f = np.nonzero(users % world == rank)[0]           # entries of my user rows
        b = np.nonzero(items % world == rank)[0]           # entries of my item rows (of A^T)
        self.a = PropGraph._single(users[f] // world, items[f], vals[f], (self.u_per, self.i_per * world), device, seg_max,
                                   col_relabel=lambda c: gathered_position(c, n_item, world))
        self.at = PropGraph._single(items[b] // world, users[b], vals[b], (self.i_per, self.u_per * world), device, seg_max,
                                    col_relabel=lambda c: gathered_position(c, n_user, world))
        self.nnz_local = int(f.size)
        self._loc = ((users[f], items[f], vals[f]), (users[b], items[b], vals[b]), seg_max)
        self._blocks = None

    @classmethod
    def from_local_entries(cls, fwd, bwd, n_user, n_item, world, rank, device, group=None, seg_max=SEG_MAX):
        """Shard-local construction (nothing global is ever materialized): `fwd` = (users, items) of the interactions of
        this rank's USER rows, `bwd` = the interactions of its ITEM rows (data_utils.synth.sharded_cells, or a loader
        that reads only those).  Values are LightGCL's 1 / sqrt(deg_u * deg_i) (lightgcl.py:17-20); every rank knows the
        degrees of its own rows, the two degree vectors are all-gathered (4 (U + I) bytes, once)."""
        self = object.__new__(cls)
        self.n_user, self.n_item, self.world, self.rank = int(n_user), int(n_item), int(world), int(rank)
        self.u_per, self.i_per = rows_per_rank(n_user, world), rows_per_rank(n_item, world)
        self.u_local = int(local_rows(n_user, world, rank).size)
        self.i_local = int(local_rows(n_item, world, rank).size)
        self.device = torch.device(device)
        (fu, fi), (bu, bi) = fwd, bwd
        deg_u_loc = np.bincount(fu // world, minlength=self.u_per).astype(np.float32)       # my users' degrees
        deg_i_loc = np.bincount(bi // world, minlength=self.i_per).astype(np.float32)       # my items' degrees
        deg_u = _all_gather_host(deg_u_loc, world, group)                                    # [rank][local] layout
        deg_i = _all_gather_host(deg_i_loc, world, group)
        du = lambda u: deg_u[gathered_position(u, n_user, world)]
        di = lambda i: deg_i[gathered_position(i, n_item, world)]
        vf = (1.0 / np.sqrt(du(fu) * di(fi))).astype(np.float32)
        vb = (1.0 / np.sqrt(du(bu) * di(bi))).astype(np.float32)
        self.a = PropGraph._single(fu // world, fi, vf, (self.u_per, self.i_per * world), device, seg_max,
                                   col_relabel=lambda c: gathered_position(c, n_item, world))
        self.at = PropGraph._single(bi // world, bu, vb, (self.i_per, self.u_per * world), device, seg_max,
                                    col_relabel=lambda c: gathered_position(c, n_user, world))
        self.nnz_local = int(fu.size)
        self._loc = ((fu, fi, vf), (bu, bi, vb), seg_max)
        self._blocks = None
        return self

    def source_blocks(self):
        """(A blocks, A^T blocks) by SOURCE rank: A block q = my user rows x the items rank q owns (local item ids), A^T block q = my item
        rows x the users rank q owns -- the operands of the pipelined exchange (`ShardedLightGCL(mode='pipelined')`: the shard of rank
        q is multiplied as soon as its broadcast has landed, the later ones still on the wire; SURVEY.md 8e "overlap").  Built on
        first use from this rank's own entries; the square case is ShardedGraph.source_blocks."""
        if self._blocks is None:
            (fu, fi, fv), (bu, bi, bv), seg_max = self._loc
            world = self.world
            a_blocks, at_blocks = [], []
            for q in range(world):
                sel = np.nonzero(fi % world == q)[0]
                a_blocks.append(PropGraph._single(fu[sel] // world, fi[sel] // world, fv[sel], (self.u_per, self.i_per), self.device, seg_max))
                sel = np.nonzero(bu % world == q)[0]
                at_blocks.append(PropGraph._single(bi[sel] // world, bu[sel] // world, bv[sel], (self.i_per, self.u_per), self.device, seg_max))
            self._blocks = (a_blocks, at_blocks)
        return self._blocks

    def local_users(self, full):
        """rows of a full [U, ...] tensor owned by this rank, zero-padded to u_per rows"""
        return _take_local(full, self.n_user, self.u_per, self.world, self.rank)

    def local_items(self, full):
        return _take_local(full, self.n_item, self.i_per, self.world, self.rank)


def _all_gather_host(x_local, world, group=None):
    """all-gather of equally sized host vectors (build-time metadata only)"""
    if solo(world):
        return x_local
    t_loc = torch.from_numpy(np.ascontiguousarray(x_local))
    dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    out = [torch.empty_like(t_loc, device=dev) for _ in range(world)]
    dist.all_gather(out, t_loc.to(dev), group=group)
    return torch.cat(out).cpu().numpy()


def _take_local(full, n, n_per, world, rank):
    ids = torch.from_numpy(local_rows(n, world, rank)).to(full.device)
    out = torch.zeros((n_per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    out[:ids.numel()] = full[ids]
    return out


class _ShardedProductFn(torch.autograd.Function):
    """y_local = M[my rows, :] @ x with x row-sharded: one all-gather + the local SpMM; the backward is the same
    with the transposed matrix's shard (dx_local = M^T[my rows, :] @ all-gather(dy_local)): no reduction crosses a GPU"""

    @staticmethod
    def forward(ctx, x_local, plan_fwd, plan_bwd, world, spmm_fn, group):
        ctx.plan_bwd, ctx.world, ctx.spmm_fn, ctx.group = plan_bwd, world, spmm_fn, group
        return spmm_fn(plan_fwd, all_gather_rows(x_local.contiguous(), world, group), None, None, True)

    @staticmethod
    def backward(ctx, gy):
        gx = ctx.spmm_fn(ctx.plan_bwd, all_gather_rows(gy.contiguous(), ctx.world, ctx.group), None, None, True)
        return gx, None, None, None, None, None


def _default_add_tables(out, a, b):
    """out = a + b over whole tables in one launch (sslrec_add_tables_f32); out may alias a or b"""
    from . import _lib
    _lib.check(_lib.load().sslrec_add_tables_f32(a.data_ptr(), b.data_ptr(), None, out.data_ptr(), out.numel(),
                                                 torch.cuda.current_stream().cuda_stream), 'sslrec_add_tables_f32')
    return out


class _ShardedLightGCLGraphViewFn(torch.autograd.Function):
    """The graph view of LightGCL on row-sharded tables as ONE autograd node: E_u^l = A E_i^(l-1), E_i^l = A^T E_u^(l-1) (no residual:
    lightgcl.py:78-79, 88-89), returns (sum_l E_u^l, sum_l E_i^l, E_u^1 .. E_u^(L-1), E_i^1 .. E_i^(L-1)) -- the layer sums (:92-93) and the
    intermediate tables the SVD view reads (:83-84).

    What it replaces: 2 L separate product nodes + 2 L table-sized stock additions forward (`sum(e_u)`, `sum(e_i)`), the same again in
    backward (9.1 ms of elementwise launches on config 5's 640 MB tables, profiles/r05/cfg5_row_sharded_step.json).  Here the layer sum
    rides in the product's epilogue (acc_out = acc_in + y, as on one GPU) and the backward chain is written by hand:
        g E_i^(l-1) = [g S_i + g E_i^(l-1) from outside] + A^T-shard . gather(g E_u^l)        (and the mirror image for users)
    with the bracket as the product's acc_in.
    mode 'all_gather': one all-gather + one product launch per table and layer (bit-identical to one rank walking the same layout);
    mode 'pipelined' : the operand travels as P broadcasts enqueued up front; the product runs source rank by source rank on
                       `source_blocks()`, own block first, every later block as its shard lands (sums grouped by source rank: equal to
                       the all-gather form to rounding, ~1e-7, not bitwise)."""

    @staticmethod
    def forward(ctx, eu0, ei0, cfg):
        sb, L, spmm, group, mode, add = cfg['sb'], cfg['L'], cfg['spmm_fn'], cfg['group'], cfg['mode'], cfg['add_fn']
        ctx.cfg = cfg
        eu0, ei0 = eu0.contiguous(), ei0.contiguous()
        if L == 0:
            return eu0.clone(), ei0.clone()
        product = _ShardedLightGCLGraphViewFn._product
        tot_u, tot_i = torch.empty_like(eu0), torch.empty_like(ei0)
        xu, xi = eu0, ei0
        mids_u, mids_i = [], []
        for l in range(1, L + 1):
            last = l == L
            zu = product(cfg, 'a', xi, eu0 if l == 1 else tot_u, tot_u, not last)          # A   . E_i^(l-1)
            zi = product(cfg, 'at', xu, ei0 if l == 1 else tot_i, tot_i, not last)         # A^T . E_u^(l-1)
            if not last:
                mids_u.append(zu)
                mids_i.append(zi)
            xu, xi = zu, zi
        return (tot_u, tot_i) + tuple(mids_u) + tuple(mids_i)

    @staticmethod
    def _product(cfg, which, x_local, acc_in, acc_out, want_y):
        """y = M[my rows, :] @ x (x row-sharded) with acc_out = acc_in + y; M = A ('a') or A^T ('at')"""
        sb, spmm, group = cfg['sb'], cfg['spmm_fn'], cfg['group']
        if cfg['mode'] != 'pipelined':
            return spmm(getattr(sb, which), all_gather_rows(x_local, sb.world, group), acc_in, acc_out, want_y)
        blocks = sb.source_blocks()[0 if which == 'a' else 1]
        y = None
        for k, (q, xq) in enumerate(shards_pipelined(x_local, sb.world, sb.rank, group)):
            if want_y:      # the product itself is wanted (next layer's operand): the blocks add up in y, the layer sum takes one more pass
                if k == 0:
                    y = spmm(blocks[q], xq, None, None, True)
                else:
                    spmm(blocks[q], xq, y, y, False)
            else:           # only the running sum is wanted: every block adds straight into it
                spmm(blocks[q], xq, acc_in if k == 0 else acc_out, acc_out, False)
        if want_y:
            cfg['add_fn'](acc_out, acc_in, y)
        return y

    @staticmethod
    def backward(ctx, g_su, g_si, *g_mids):
        cfg = ctx.cfg
        sb, L, add = cfg['sb'], cfg['L'], cfg['add_fn']
        if L == 0:
            return g_su, g_si, None
        product = _ShardedLightGCLGraphViewFn._product
        zero_u = zero_i = None
        if g_su is None:
            g_su = zero_u = torch.zeros((sb.u_per, g_si.shape[1] if g_si is not None else g_mids[0].shape[1]), device=sb.device)
        if g_si is None:
            g_si = zero_i = torch.zeros((sb.i_per, g_su.shape[1]), device=sb.device)
        g_su, g_si = g_su.contiguous(), g_si.contiguous()
        gm_u, gm_i = g_mids[:L - 1], g_mids[L - 1:]
        gu, gi = g_su, g_si                                  # gradients of E_u^L, E_i^L
        for l in range(L, 0, -1):
            # what reaches E^(l-1) directly: the layer sum's gradient (+ the outside use of an intermediate table, 1 <= l-1 <= L-1)
            cu, ci = g_su, g_si
            if l - 1 >= 1:
                if gm_u[l - 2] is not None:
                    cu = add(torch.empty_like(g_su), g_su, gm_u[l - 2].contiguous())
                if gm_i[l - 2] is not None:
                    ci = add(torch.empty_like(g_si), g_si, gm_i[l - 2].contiguous())
            nu, ni = torch.empty_like(g_su), torch.empty_like(g_si)
            product(cfg, 'a', gi, cu, nu, False)             # g E_u^(l-1) = cu + A   . g E_i^l
            product(cfg, 'at', gu, ci, ni, False)            # g E_i^(l-1) = ci + A^T . g E_u^l
            gu, gi = nu, ni
        return gu, gi, None


def _default_rankq(left_local, right_local, x_local, reduce):
    """left_local @ reduce(right_local @ x_local) with the rank-q streaming kernels (ops.lowrank_*); reduce = in-place
    sum over the ranks of a [q, d] tensor"""
    return _ShardedLowRankFn.apply(x_local, left_local, right_local, reduce)


class _ShardedLowRankFn(torch.autograd.Function):
    """LightGCL's SVD view `u_mul_s @ (vt @ E)` (lightgcl.py:83-84) with E, vt's columns and u_mul_s's rows sharded:
    partial [q, d] products are all-reduced (q*d floats), forward and backward"""

    @staticmethod
    def forward(ctx, x_local, left_local, right_local, reduce):
        s_ = ops.rankq_reduce(right_local, False, x_local.contiguous())
        reduce(s_)
        ctx.save_for_backward(left_local, right_local)
        ctx.reduce = reduce
        return ops.rankq_expand(left_local, True, s_)

    @staticmethod
    def backward(ctx, gy):
        left_local, right_local = ctx.saved_tensors
        s_ = ops.rankq_reduce(left_local, True, gy.contiguous())
        ctx.reduce(s_)
        return ops.rankq_expand(right_local, False, s_), None, None, None


class _ShardedSvdViewFn(torch.autograd.Function):
    """One table of LightGCL's SVD view on row-sharded tables, all layers at once: G = E^0 + sum_l left (right E_other^l) (lightgcl.py:83-84,
    94-95) = E^0 + left (sum_l right E_other^l) -- the rank-q map is linear, so the L partial [q, d] products are added BEFORE the one
    all-reduce and the one expansion: per table 1 expansion + 1 table addition instead of L + L, and in backward ONE reduce / all-reduce /
    expand whose result is every layer's gradient (the upstream gradient of G is the same for all of them).  Equal to the per-layer form
    to rounding (the q x d sums are formed in a different order), not bitwise."""

    @staticmethod
    def forward(ctx, e0_self, left_local, right_local, fns, *xs):
        reduce_fn, expand_fn, allreduce, add_fn = fns
        s_ = None
        for x in xs:
            part = reduce_fn(right_local, False, x.contiguous())
            s_ = part if s_ is None else s_.add_(part)          # [q, d]: a few hundred numbers
        allreduce(s_)
        ctx.save_for_backward(left_local, right_local)
        ctx.fns, ctx.n_x = fns, len(xs)
        y = expand_fn(left_local, True, s_)
        return add_fn(y, e0_self.contiguous(), y)

    @staticmethod
    def backward(ctx, g):
        left_local, right_local = ctx.saved_tensors
        reduce_fn, expand_fn, allreduce, _ = ctx.fns
        t_ = reduce_fn(left_local, True, g.contiguous())
        allreduce(t_)
        r = expand_fn(right_local, False, t_)
        return (g, None, None, None) + (r,) * ctx.n_x


class ShardedLightGCL(torch.nn.Module):
    """LightGCL (reference models/general_cf/lightgcl.py:73-125) with both embedding tables ROW-SHARDED over the ranks.

    Per layer: two sharded products (A by user rows, A^T by item rows: one all-gather + one local SpMM each, and the
    mirror image in backward) and the rank-q SVD view with a q x d all-reduce.  Losses: LightGCL's BPR on rows exchanged
    with one small all-reduce, its un-normalized InfoNCE with `all` kept sharded (variant 1 of the staged
    sslrec_infonce_shard_* ABI: B row sums / B x d anchor gradients all-reduced), the regularizer on the local rows
    (the replicated d x d `Ws` count once, on rank 0).  Loss values are identical on every rank except `reg_local`.
    `factors` = this rank's slices of the SVD factors: (ut [q, u_per], vt [q, i_per], u_mul_s [u_per, q],
    v_mul_s [i_per, q]), zero in the padding positions."""

    def __init__(self, sb, init_users, init_items, factors, layer_num, temp, spmm_fn=None, rankq_fn=None, group=None, mode='all_gather',
                 add_fn=None, lowrank_ops=None):
        """mode: 'all_gather' (one all-gather + one product launch per table and layer) | 'pipelined' (per-source-rank broadcasts
        overlapped with per-source-rank block products) | 'separate' (rounds 4-5: one autograd node per product, layer sums by
        stock additions -- kept as the statement the fused node is tested against).  add_fn(out, a, b): table addition (tests inject
        a CPU one).  lowrank_ops = (reduce(m, transposed, x), expand(m, transposed, s)): the two rank-q kernels of the fused SVD view
        (_ShardedSvdViewFn); default ops.rankq_reduce / rankq_expand -- unless `rankq_fn` is injected without them (tests of the per-layer form)."""
        super().__init__()
        if lowrank_ops is None and rankq_fn is None:
            lowrank_ops = (ops.rankq_reduce, ops.rankq_expand)
        self.lowrank_ops = lowrank_ops
        if mode not in ('all_gather', 'pipelined', 'separate'):
            raise ValueError("mode %r: 'all_gather', 'pipelined' or 'separate'" % (mode,))
        self.sb, self.layer_num, self.temp, self.mode = sb, int(layer_num), float(temp), mode
        self.spmm_fn, self.rankq_fn, self.group = spmm_fn or _default_spmm, rankq_fn or _default_rankq, group
        self.add_fn = add_fn or _default_add_tables
        self.local_user_embeds = torch.nn.Parameter(sb.local_users(init_users).to(sb.device))
        self.local_item_embeds = torch.nn.Parameter(sb.local_items(init_items).to(sb.device))
        self.ut, self.vt, self.u_mul_s, self.v_mul_s = (f.to(sb.device).contiguous() for f in factors)
        self.last_parts = {}

    def _reduce(self, t):
        if not solo(self.sb.world):
            all_reduce_sum(t, self.group)
        return t

    def forward(self):
        """local rows of (E_u, E_i, G_u, G_i): the layer sums of the graph view and of the SVD view (lightgcl.py:76-95)"""
        sb, fn = self.sb, _ShardedProductFn.apply
        if self.mode != 'separate':
            # the graph view as one node (layer sums in the products' epilogues, hand-written backward chain, optionally the
            # pipelined exchange); the SVD view reads its intermediate tables
            L = self.layer_num
            cfg = {'sb': sb, 'L': L, 'spmm_fn': self.spmm_fn, 'group': self.group, 'mode': self.mode, 'add_fn': self.add_fn}
            outs = _ShardedLightGCLGraphViewFn.apply(self.local_user_embeds, self.local_item_embeds, cfg)
            sum_u, sum_i = outs[0], outs[1]
            lay_u = [self.local_user_embeds] + list(outs[2:2 + max(L - 1, 0)])
            lay_i = [self.local_item_embeds] + list(outs[2 + max(L - 1, 0):])
            if self.lowrank_ops is not None and L >= 1:      # the SVD view of a table as one node: one expansion, one table addition
                fns = (self.lowrank_ops[0], self.lowrank_ops[1], self._reduce, self.add_fn)
                g_u = _ShardedSvdViewFn.apply(self.local_user_embeds, self.u_mul_s, self.vt, fns, *lay_i)
                g_i = _ShardedSvdViewFn.apply(self.local_item_embeds, self.v_mul_s, self.ut, fns, *lay_u)
                return sum_u, sum_i, g_u, g_i
            g_u, g_i = self.local_user_embeds, self.local_item_embeds
            for l in range(L):
                g_u = g_u + self.rankq_fn(self.u_mul_s, self.vt, lay_i[l], self._reduce)
                g_i = g_i + self.rankq_fn(self.v_mul_s, self.ut, lay_u[l], self._reduce)
            return sum_u, sum_i, g_u, g_i
        e_u, e_i = [self.local_user_embeds], [self.local_item_embeds]
        g_u, g_i = [self.local_user_embeds], [self.local_item_embeds]
        for _ in range(self.layer_num):
            z_u = fn(e_i[-1], sb.a, sb.at, sb.world, self.spmm_fn, self.group)          # A   @ E_i
            z_i = fn(e_u[-1], sb.at, sb.a, sb.world, self.spmm_fn, self.group)          # A^T @ E_u
            g_u.append(self.rankq_fn(self.u_mul_s, self.vt, e_i[-1], self._reduce))
            g_i.append(self.rankq_fn(self.v_mul_s, self.ut, e_u[-1], self._reduce))
            e_u.append(z_u)
            e_i.append(z_i)
        return sum(e_u), sum(e_i), sum(g_u), sum(g_i)

    def _rows(self, s_local, ids):
        return _ExchangeRowsFn.apply(s_local, ids, self.sb.world, self.sb.rank, self.group)

    def _infonce(self, e1, e2, all_local, infonce_fn):
        if infonce_fn is not None:
            return infonce_fn(e1, e2, all_local, self.temp)
        return ops.infonce_loss_sharded(e1, e2, all_local, self.temp, 1, self._reduce)

    def lightgcl_loss(self, batch, cl_weight, reg_weight, extra_params=(), bpr_fn=None, reg_fn=None, infonce_fn=None):
        """bpr + cl_weight * cl + reg_weight * (this rank's share of the regularizer); reference lightgcl.py:99-125"""
        ancs, poss, negs = batch[:3]
        B = ancs.shape[0]
        sb = self.sb
        e_u, e_i, g_u, g_i = self.forward()
        anc, gu_a = self._rows(e_u, ancs), self._rows(g_u, ancs)
        pn = self._rows(e_i, torch.cat([poss, negs]))
        gi_p = self._rows(g_i, poss)
        bpr = (bpr_fn or (lambda a, p, n: ops.bpr_loss(a, p, n, variant=1)))(anc, pn[:B], pn[B:]) / B
        cl = (self._infonce(gu_a, anc, e_u[:sb.u_local], infonce_fn) +
              self._infonce(gi_p, pn[:B], e_i[:sb.i_local], infonce_fn)) / B
        sq = reg_fn or ops.sum_squares
        reg = sq(self.local_user_embeds) + sq(self.local_item_embeds)
        if sb.rank == 0:
            for w in extra_params:
                reg = reg + sq(w)
        self.last_parts = {'bpr_loss': bpr.detach(), 'cl_loss': (cl_weight * cl).detach(), 'reg_local': reg.detach()}
        return bpr + cl_weight * cl + reg_weight * reg
