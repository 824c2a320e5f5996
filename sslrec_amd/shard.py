"""Row-sharded propagation over the GPUs of one node (one process per GPU, RCCL over xGMI
through torch.distributed; backend "nccl" IS RCCL on ROCm).

The reference has no multi-GPU code at all; this is the §8(e) design:

  * the N = U+I rows of the embedding table are dealt CYCLICALLY to the P ranks
    (row r lives on rank r % P at local index r // P), which balances the power-law
    degrees without any statistics; every rank keeps the CSR of its rows of A (and of
    A^T for the backward pass -- the same arrays when A is symmetric) with GLOBAL
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


def gathered_position(ids, n, world):
    """position of global row ids inside the all-gathered [rank][local] layout"""
    ids = np.asarray(ids, dtype=np.int64)
    return (ids % world) * rows_per_rank(n, world) + ids // world


class ShardedGraph:
    """This rank's slice of a square adjacency given as global COO (rows, cols, vals)."""

    def __init__(self, rows, cols, vals, n, world, rank, device, seg_max=SEG_MAX):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float32)
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        self.n_per = rows_per_rank(n, world)
        self.n_local = int(local_rows(n, world, rank).size)
        self.device = torch.device(device)
        n_gathered = self.n_per * world
        # forward shard: rows of A owned by this rank, columns in gathered layout
        self.coo_ids_fwd = np.nonzero(rows % world == rank)[0]
        self.coo_ids_bwd = np.nonzero(cols % world == rank)[0]
        f, b = self.coo_ids_fwd, self.coo_ids_bwd
        # PropGraph(fwd = A_shard [n_per x n_gathered]); its own .bwd (transpose of the shard) is unused:
        # the backward pass needs ROWS of A^T, i.e. a second forward-type plan.
        self.a = PropGraph._single(rows[f] // world, cols[f], vals[f], (self.n_per, n_gathered), device, seg_max,
                                   col_relabel=lambda c: gathered_position(c, n, world))
        self.at = PropGraph._single(cols[b] // world, rows[b], vals[b], (self.n_per, n_gathered), device, seg_max,
                                    col_relabel=lambda c: gathered_position(c, n, world), share_from=self.a)
        self.nnz_local = int(f.size)
        self._col_sharded = None
        self._coo = (rows, cols, vals, seg_max)

    def col_sharded(self):
        """(A[:, my cols], A^T[:, my cols]) with rows re-labelled into the [rank][local] layout -- the
        operands of the reduce-scatter formulation; built on first use."""
        if self._col_sharded is None:
            rows, cols, vals, seg_max = self._coo
            n, world, rank = self.n, self.world, self.rank
            n_gathered = self.n_per * world
            f = np.nonzero(cols % world == rank)[0]            # entries whose COLUMN this rank owns
            b = np.nonzero(rows % world == rank)[0]
            a_c = PropGraph._single(gathered_position(rows[f], n, world), cols[f] // world, vals[f],
                                    (n_gathered, self.n_per), self.device, seg_max)
            at_c = PropGraph._single(gathered_position(cols[b], n, world), rows[b] // world, vals[b],
                                     (n_gathered, self.n_per), self.device, seg_max, share_from=a_c)
            self._col_sharded = (a_c, at_c)
        return self._col_sharded

    def to_local(self, full):
        """rows of a full [N, d] host/device tensor owned by this rank, padded to n_per rows"""
        ids = torch.from_numpy(local_rows(self.n, self.world, self.rank)).to(full.device)
        out = torch.zeros((self.n_per, full.shape[1]), dtype=full.dtype, device=full.device)
        out[:ids.numel()] = full[ids]
        return out


def all_gather_rows(x_local, world, group=None):
    """[n_per, d] per rank -> [world * n_per, d] in [rank][local] order (one collective)"""
    if world == 1:
        return x_local
    out = torch.empty((world * x_local.shape[0], x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
    return out


def reduce_scatter_rows(y_full, world, group=None):
    """[world * n_per, d] partial results per rank -> this rank's [n_per, d] rows of their sum"""
    if world == 1:
        return y_full
    out = torch.empty((y_full.shape[0] // world, y_full.shape[1]), dtype=y_full.dtype, device=y_full.device)
    dist.reduce_scatter_tensor(out, y_full.contiguous(), op=dist.ReduceOp.SUM, group=group)
    return out


def _default_spmm(plan_graph, x_gathered, acc_in, acc_out, want_y):
    return ops.spmm_raw(plan_graph, x_gathered, 'fwd', acc_in=acc_in, acc_out=acc_out, want_y=want_y)


class _ShardedPropagateSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e0_local, sg, layer_num, spmm_fn, group):
        ctx.sg, ctx.layer_num, ctx.spmm_fn, ctx.group = sg, layer_num, spmm_fn, group
        e0_local = e0_local.contiguous()
        if layer_num == 0:
            return e0_local.clone()
        total = torch.empty_like(e0_local)
        x = e0_local
        for l in range(layer_num):
            xg = all_gather_rows(x, sg.world, group)
            last = (l == layer_num - 1)
            x = spmm_fn(sg.a, xg, e0_local if l == 0 else total, total, not last)
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


def sharded_propagate_sum(sg, e0_local, layer_num, spmm_fn=None, group=None, mode='all_gather'):
    """Local rows of  E0 + sum_l A^l E0  for a row-sharded table (differentiable).
    mode: 'all_gather' (row-sharded A, bit-identical to one GPU) or 'reduce_scatter' (column-sharded A)."""
    fn = {'all_gather': _ShardedPropagateSumFn, 'reduce_scatter': _ShardedPropagateSumRsFn}[mode]
    return fn.apply(e0_local, sg, int(layer_num), spmm_fn or _default_spmm, group)


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


class ShardedGraphCF(torch.nn.Module):
    """LightGCN-family model whose stacked embedding table [users; items] is ROW-SHARDED over the
    ranks (parameter = this rank's rows, so optimizer state is sharded too).

    One training step = sharded propagation (one all-gather per layer, see above) -> ONE more
    all-gather of the final embeddings -> every rank evaluates the losses for ITS SLICE of the
    batch (anchors b with b % P == rank) against the full tables with the single-GPU fused
    kernels -> autograd's backward of the all-gather is a reduce-scatter that hands every rank the
    summed gradient of its rows -> sharded backward propagation.  The losses therefore scale 1/P
    in compute (the InfoNCE B x M product is split over anchors) at the price of two extra
    collectives of N*d*4 bytes per step; the loss VALUE a rank returns is its slice's share --
    `dist.all_reduce` it for logging.
    """

    def __init__(self, sg, n_user, n_item, init_table, layer_num, spmm_fn=None, group=None, mode='all_gather'):
        super().__init__()
        self.sg, self.n_user, self.n_item, self.layer_num = sg, int(n_user), int(n_item), int(layer_num)
        self.spmm_fn, self.group, self.mode = spmm_fn, group, mode
        self.local_embeds = torch.nn.Parameter(sg.to_local(init_table).to(sg.device))
        pos = gathered_position(np.arange(sg.n), sg.n, sg.world)
        self.register_buffer('pos_users', torch.from_numpy(pos[:self.n_user]).to(sg.device), persistent=False)
        self.register_buffer('pos_items', torch.from_numpy(pos[self.n_user:]).to(sg.device), persistent=False)

    def tables(self):
        """(user table [U,d], item table [I,d]) of the propagated + layer-summed embeddings, full
        and in global row order on every rank, differentiable w.r.t. the local parameter rows"""
        s_local = sharded_propagate_sum(self.sg, self.local_embeds, self.layer_num, self.spmm_fn, self.group, self.mode)
        s_all = _AllGatherRowsFn.apply(s_local, self.sg.world, self.group)
        return s_all.index_select(0, self.pos_users), s_all.index_select(0, self.pos_items)

    def batch_slice(self, batch):
        """this rank's share of a batch of index tensors (anchors dealt cyclically)"""
        return [t[self.sg.rank::self.sg.world] for t in batch]

    def reg_loss(self, reg_fn=None):
        """sum of squares of the LOCAL rows (padding rows are zero and stay zero)"""
        return (reg_fn or ops.sum_squares)(self.local_embeds)

    def lightgcn_loss(self, batch, reg_weight, bpr_fn=None, reg_fn=None):
        """this rank's share of LightGCN's loss (reference lightgcn.py:45-56); summing the returned
        value over ranks gives the single-GPU loss"""
        bpr_fn = bpr_fn or (lambda u, i, a, p, n: ops.bpr_loss_gathered(u, i, a, p, n, 0))
        users, items = self.tables()
        ancs, poss, negs = self.batch_slice(batch)
        bpr = bpr_fn(users, items, ancs, poss, negs) / batch[0].shape[0]
        return bpr + reg_weight * self.reg_loss(reg_fn)
