"""Feature-sliced propagation over the GPUs of one node: every rank holds ALL rows of the embedding tables but only
d / P of their columns, and the whole adjacency.

Why (measurements: EXPERIMENTS.md B §5.0): the propagation `E_l = A E_{l-1}` (reference models/general_cf/lightgcn.py:28-43) acts on every
embedding COLUMN independently, so a column slice of the layer sum needs nothing from the other slices -- the L
products of a step, forward and backward, run without a single collective.  Row-sharded tables (sslrec_amd/shard.py,
the formulation BASELINE.json words) pay one table-sized all-gather per layer and direction; on xGMI's point-to-point
links that exchange is several times the local SpMM at every size we measured or modelled (amazon-book: 37 MB per
layer against a 80 us product; config 5: 4.5 GB per exchange against 3 ms).  Row sharding buys memory capacity, and
an MI355X does not need it here: with 288 GB of HBM a GPU holds the whole adjacency of config 5 (5 GB) next to its
slice of the tables.  What remains on the wire:

  * the 3B batch rows of a step, whose dot products need all d columns: every rank gathers its [3B, d/P] slice and
    ONE small all-gather assembles [3B, d] (393 KB per rank at B = 4096, d = 64, P = 8) -- `rows()`.  Every rank then
    evaluates the same batch loss, so the backward of the exchange is local: a rank keeps the columns it owns;
  * for the contrastive terms (cal_infonce_loss, models/loss_utils.py:30-39) the score of a pair needs all d
    columns of BOTH rows, so the two final tables are transposed once per step from column slices to row shards by
    an all-to-all (`to_row_shards`, each rank sends (P-1)/P of its slice: 1/P of a table per rank instead of the
    (P-1)/P of a table PER LAYER of the row-sharded all-gather), and the row-sharded InfoNCE of shard.py takes over.

Numerics: a slice of the result is computed by the same kernel in the same per-row entry order as on one GPU; only
the cut of heavy rows into chunks depends on the layout's width, so slices agree with the single-GPU table to ~1e-7,
not bitwise.  Parameters and optimizer state are sliced like the tables.

`propagate_fn`, `bpr_fn`, `reg_fn`, `scatter_fn` are injectable so that the partition / collective logic runs on CPU
under gloo (tests/test_shard_gloo.py); the defaults are the HIP ops.
"""
import torch
import torch.distributed as dist

from . import ops
from .shard import all_gather_rows, all_reduce_sum, solo

# widths the column-swept SpMM is instantiated for (spmm_swept.hip); 8 and 16 exist for this module
SLICE_WIDTHS = (8, 16, 32, 64, 128, 256)


def slice_bounds(d, world, rank):
    """[lo, hi) = the embedding columns rank `rank` owns"""
    if d % world != 0:
        raise ValueError('embedding size %d is not divisible by the %d ranks' % (d, world))
    w = d // world
    return rank * w, (rank + 1) * w


def _default_scatter(src, ids, n_rows):
    """zeros [n_rows, w] with src[k] added to row ids[k], duplicates in a fixed order (sslrec_scatter_add_rows_f32)"""
    from . import _lib
    lib = _lib.load()
    src = src.contiguous()
    K, w = src.shape
    out = torch.zeros((n_rows, w), dtype=torch.float32, device=src.device)
    ws = torch.empty(lib.sslrec_scatter_ws_bytes(K) // 4 + 1, dtype=torch.float32, device=src.device)
    rc = lib.sslrec_scatter_add_rows_f32(src.data_ptr(), ids.data_ptr(), K, w, out.data_ptr(), ws.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, 'sslrec_scatter_add_rows_f32')
    return out


def _host_scatter(src, ids, n_rows):
    out = torch.zeros((n_rows, src.shape[1]), dtype=src.dtype, device=src.device)
    out.index_add_(0, ids, src)
    return out


class _GatherColumnsFn(torch.autograd.Function):
    """rows `ids` of a feature-sliced table [N, w] as full-width rows [K, P*w], identical on every rank: one all-gather of
    the [K, w] slices.  Backward: the caller evaluates the SAME loss on every rank, so the incoming gradient is already
    the full one -- a rank keeps its own columns and scatters them into its [N, w] table (no collective)."""

    @staticmethod
    def forward(ctx, s_local, ids, world, rank, group, scatter_fn):
        mine = s_local.index_select(0, ids)
        ctx.save_for_backward(ids)
        ctx.meta = (s_local.shape[0], s_local.shape[1], world, rank, scatter_fn)
        if solo(world):
            return mine
        K, w = mine.shape
        full = all_gather_rows(mine, world, group)              # [world * K, w], rank-major
        return full.view(world, K, w).permute(1, 0, 2).reshape(K, world * w)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        n_rows, w, world, rank, scatter_fn = ctx.meta
        gs = g[:, rank * w:(rank + 1) * w].contiguous()
        return scatter_fn(gs, ids, n_rows), None, None, None, None, None


def _all_to_all(send, recv_rows, group):
    """send[q] -> rank q; returns what every rank sent here (recv_rows[p] rows from rank p, known to both sides).  RCCL: one
    all_to_all; gloo (tests only: host memory, no all-to-all with uneven splits on every build): one broadcast per
    (source, destination) pair"""
    world = len(send)
    if solo(world):
        return [send[0]]
    rank = dist.get_rank(group)
    w, dt, dev = send[0].shape[1], send[0].dtype, send[0].device
    if dist.get_backend(group) == 'gloo':
        g_of = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
        rows_of = torch.tensor([t.shape[0] for t in send], dtype=torch.int64)
        recv = [None] * world
        for src in range(world):
            cnt = rows_of.clone() if src == rank else torch.zeros(world, dtype=torch.int64)
            dist.broadcast(cnt, src=g_of(src), group=group)
            for dst in range(world):
                buf = send[dst].detach().cpu().contiguous() if src == rank else torch.empty((int(cnt[dst]), w), dtype=dt)
                dist.broadcast(buf, src=g_of(src), group=group)
                if dst == rank:
                    recv[src] = buf.to(dev)
        return recv
    recv = [torch.empty((int(n), w), dtype=dt, device=dev) for n in recv_rows]
    dist.all_to_all(recv, [t.contiguous() for t in send], group=group)
    return recv


def row_block(m, world, rank):
    """[lo, hi) = the rows of an m-row table rank `rank` owns after the transposition (contiguous, near-equal blocks)"""
    base, rem = divmod(m, world)              # the first `rem` ranks hold one row more: no rank is empty unless m < world
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _SlicesToRowsFn(torch.autograd.Function):
    """[M, w] column slice of a table on every rank -> this rank's ROW block [m_r, P*w] of the full-width table: one
    all-to-all (rank r sends rank q the rows of q's block).  Backward is the same exchange reversed."""

    @staticmethod
    def forward(ctx, x_slice, world, rank, group):
        M, w = x_slice.shape
        ctx.meta = (M, w, world, rank, group)
        if solo(world):
            return x_slice
        send = [x_slice[slice(*row_block(M, world, q))] for q in range(world)]
        lo, hi = row_block(M, world, rank)
        recv = _all_to_all(send, [hi - lo] * world, group)      # recv[p] = columns of rank p, my rows
        return torch.cat(recv, dim=1).contiguous()

    @staticmethod
    def backward(ctx, g_rows):
        M, w, world, rank, group = ctx.meta
        if solo(world):
            return g_rows, None, None, None
        send = [g_rows[:, p * w:(p + 1) * w].contiguous() for p in range(world)]
        blocks = [row_block(M, world, q) for q in range(world)]
        recv = _all_to_all(send, [b - a for a, b in blocks], group)      # recv[q] = my columns of rank q's rows
        return torch.cat(recv, dim=0).contiguous(), None, None, None


def to_row_shards(x_slice, world, rank, group=None):
    """column slices -> row blocks (differentiable); the operand layout of ops.infonce_loss_sharded"""
    return _SlicesToRowsFn.apply(x_slice, world, rank, group)


class FeatureSlicedGraphCF(torch.nn.Module):
    """LightGCN-family model whose stacked table [users; items] is sliced by embedding COLUMN over the ranks
    (parameter = this rank's [N, d/P] slice; optimizer state is sliced with it).  `graph` is the WHOLE adjacency
    (sslrec_amd.graph.PropGraph, or whatever `propagate_fn(graph, e0, layer_num)` accepts).

    Step = local L-layer propagation of the slice (no collective) -> `rows()`: the 3B batch rows assembled by one small
    all-gather -> the batch loss evaluated identically on every rank -> local backward.  Loss values: the batch terms are
    identical on all ranks, the regularizer is this rank's share (`last_parts['reg_local']`; all-reduce it for logging).
    """

    def __init__(self, graph, n_user, n_item, init_table, layer_num, world, rank, group=None, device=None,
                 propagate_fn=None, scatter_fn=None):
        super().__init__()
        self.graph, self.n_user, self.n_item, self.layer_num = graph, int(n_user), int(n_item), int(layer_num)
        self.world, self.rank, self.group = int(world), int(rank), group
        self.d = int(init_table.shape[1])
        self.lo, self.hi = slice_bounds(self.d, self.world, self.rank)
        self.width = self.hi - self.lo
        device = torch.device(device if device is not None else getattr(graph, 'device', 'cpu'))
        if propagate_fn is None and self.width not in SLICE_WIDTHS:
            raise ValueError('a slice of %d columns (d = %d over %d ranks) is not a width of the column-swept SpMM %s'
                             % (self.width, self.d, self.world, SLICE_WIDTHS))
        self.local_embeds = torch.nn.Parameter(init_table[:, self.lo:self.hi].detach().to(device).contiguous())
        self.propagate_fn = propagate_fn or ops.propagate_sum
        self.scatter_fn = scatter_fn or (_default_scatter if device.type == 'cuda' else _host_scatter)
        self.last_parts = {}

    # ---- propagation: purely local -------------------------------------------------------------------------------------
    def propagate(self, adj=None):
        """this rank's columns of the layer-summed propagated table [N, d/P] (differentiable); `adj` = an edge-dropped
        view of the graph (same mask on every rank: the reference's host draw under a common seed, or the Philox bit of
        the COO entry id) or None for the graph itself"""
        return self.propagate_fn(self.graph if adj is None else adj, self.local_embeds, self.layer_num)

    def noise_row_sumsq(self, noise, row_sumsq_fn=None):
        """squared L2 norm of every FULL row of one EmbedPerturb draw (aug_utils.py:130 normalizes over all d columns, a rank
        holds d/P of them).  `noise` = this rank's columns [N, d/P] of a draw every rank knows its own columns of (parity mode:
        the reference's CPU draw under a common seed) -> the ranks' partial sums are all-reduced ([N] floats); or an
        rng.PhiloxNoise token for the FULL [N, d] table (perf mode) -> every rank computes all d draws of a row, no collective"""
        if torch.is_tensor(noise):
            ss = (row_sumsq_fn or ops.row_sumsq)(noise)
            if not solo(self.world):
                all_reduce_sum(ss, self.group)
            return ss
        return ops.philox_row_sumsq(noise)

    def propagate_perturbed(self, noises, eps, row_sumsq_fn=None):
        """this rank's columns of SimGCL's perturbed propagation (simgcl.py:20-30): per layer y += eps * sign(y) * n / |n| with
        the norm over the full row; `noises` = L draws (column slices as tensors, or PhiloxNoise tokens of the full table whose
        element index keeps a slice of the result equal to the same columns of the one-GPU result)"""
        sumsq = [self.noise_row_sumsq(nz, row_sumsq_fn) for nz in noises]
        tokens = not torch.is_tensor(noises[0])
        return self.propagate_fn(self.graph, self.local_embeds, self.layer_num, noises, eps, noise_sumsq=sumsq,
                                 noise_geom=(self.d, self.lo) if tokens else None)

    def rows(self, s_local, ids):
        """full-width rows [K, d] of the stacked table for stacked ids, identical on every rank (one small all-gather)"""
        return _GatherColumnsFn.apply(s_local, ids, self.world, self.rank, self.group, self.scatter_fn)

    def batch_ids(self, batch):
        ancs, poss, negs = batch[:3]
        return torch.cat([ancs, poss + self.n_user, negs + self.n_user])

    def batch_rows(self, s_local, batch):
        B = batch[0].shape[0]
        buf = self.rows(s_local, self.batch_ids(batch))
        return buf[:B], buf[B:2 * B], buf[2 * B:]

    def full_tables(self, s_local=None):
        """(user table [U, d], item table [I, d]) on every rank, for evaluation (not differentiable): one table-sized
        all-gather along the embedding dimension"""
        with torch.no_grad():
            s = self.propagate() if s_local is None else s_local
            if not solo(self.world):
                full = all_gather_rows(s.contiguous(), self.world, self.group)          # [P * N, w], rank-major
                s = full.view(self.world, s.shape[0], s.shape[1]).permute(1, 0, 2).reshape(s.shape[0], self.d)
            return s[:self.n_user], s[self.n_user:]

    # ---- losses ----------------------------------------------------------------------------------------------------------
    def reg_loss(self, reg_fn=None):
        """sum of squares of the local columns; the ranks' shares add up to the reference's reg_params term"""
        return (reg_fn or ops.sum_squares)(self.local_embeds)

    def infonce(self, e1, e2, s_slice, temp, infonce_fn=None):
        """cal_infonce_loss(e1, e2, all, temp) (loss_utils.py:30-39) with `all` = the table whose column slice is s_slice:
        transposed to row blocks by one all-to-all, then the row-sharded kernels (B row sums / B x d anchor gradients
        all-reduced)"""
        all_local = to_row_shards(s_slice, self.world, self.rank, self.group)
        if infonce_fn is not None:
            return infonce_fn(e1, e2, all_local, temp)
        grp = self.group
        red = (lambda t: t) if solo(self.world) else (lambda t: all_reduce_sum(t, grp))
        return ops.infonce_loss_sharded(e1, e2, all_local, temp, 0, red)

    def lightgcn_loss(self, batch, reg_weight, bpr_fn=None, reg_fn=None, adj=None):
        """LightGCN's loss (reference lightgcn.py:45-56): bpr / B (same on every rank) + reg_weight * (this rank's
        share of the regularizer).  `adj`: the edge-dropped view of this step when keep_rate < 1 (lightgcn.py:33-34; the
        same on every rank), None = the graph itself"""
        B = batch[0].shape[0]
        anc, pos, neg = self.batch_rows(self.propagate(adj), batch)
        bpr = (bpr_fn(anc, pos, neg) / B) if bpr_fn is not None else ops.bpr_loss(anc, pos, neg, divisor=B)
        reg = self.reg_loss(reg_fn)
        self.last_parts = {'bpr_loss': bpr.detach(), 'reg_local': reg.detach()}
        return bpr + reg_weight * reg

    def sgl_loss(self, batch, view1, view2, reg_weight, cl_weight, temp, bpr_fn=None, reg_fn=None, infonce_fn=None):
        """SGL-ED's loss (reference sgl.py:45-65): view1 / view2 = two edge-dropped views of the graph (identical on every
        rank), BPR on the clean propagation, three InfoNCE terms between the views' batch rows against all users / items"""
        ancs, poss, negs = batch[:3]
        B = ancs.shape[0]
        v1, v2 = self.propagate(view1), self.propagate(view2)
        anc, pos, neg = self.batch_rows(self.propagate(), batch)
        bpr = (bpr_fn(anc, pos, neg) / B) if bpr_fn is not None else ops.bpr_loss(anc, pos, neg, divisor=B)
        ids = self.batch_ids(batch)
        r1, r2 = self.rows(v1, ids), self.rows(v2, ids)
        users2, items2 = v2[:self.n_user], v2[self.n_user:]
        cl = self.infonce(r1[:B], r2[:B], users2, temp, infonce_fn) + \
            self.infonce(r1[B:], r2[B:], items2, temp, infonce_fn)          # positives and negatives: one call, same `all` (sgl.py:58-59)
        cl = cl / B
        reg = self.reg_loss(reg_fn)
        self.last_parts = {'bpr_loss': bpr.detach(), 'cl_loss': cl.detach(), 'reg_local': reg.detach()}
        return bpr + reg_weight * reg + cl_weight * cl

    def simgcl_loss(self, batch, noises1, noises2, eps, reg_weight, cl_weight, temp, bpr_fn=None, reg_fn=None, infonce_fn=None,
                    row_sumsq_fn=None):
        """SimGCL's loss (reference simgcl.py:39-55): two perturbed propagations (noises*: L draws each, see propagate_perturbed)
        and the clean one -- all three without a collective in the products --, BPR on the clean view's batch rows, InfoNCE
        between the perturbed views' batch rows against all users / items through the transposition to row blocks"""
        ancs, poss = batch[0], batch[1]
        B = ancs.shape[0]
        if self.propagate_fn is ops.propagate_sum:      # the three views share ONE backward chain (ops.propagate_sum_views)
            sumsq = [[self.noise_row_sumsq(nz, row_sumsq_fn) for nz in noises] for noises in (noises1, noises2)]
            tokens = not torch.is_tensor(noises1[0])
            v1, v2, v3 = ops.propagate_sum_views(self.graph, self.local_embeds, self.layer_num, [noises1, noises2, None], eps,
                                                 noise_sumsq_views=sumsq + [None], noise_geom=(self.d, self.lo) if tokens else None)
        else:                                           # (a stand-in propagation, e.g. the CPU tests')
            v1 = self.propagate_perturbed(noises1, eps, row_sumsq_fn)
            v2 = self.propagate_perturbed(noises2, eps, row_sumsq_fn)
            v3 = self.propagate()
        anc, pos, neg = self.batch_rows(v3, batch)
        bpr = (bpr_fn(anc, pos, neg) / B) if bpr_fn is not None else ops.bpr_loss(anc, pos, neg, divisor=B)
        ids = torch.cat([ancs, poss + self.n_user])
        r1, r2 = self.rows(v1, ids), self.rows(v2, ids)
        cl = self.infonce(r1[:B], r2[:B], v2[:self.n_user], temp, infonce_fn) + \
            self.infonce(r1[B:], r2[B:], v2[self.n_user:], temp, infonce_fn)
        cl = cl / B
        reg = self.reg_loss(reg_fn)
        self.last_parts = {'bpr_loss': bpr.detach(), 'cl_loss': cl.detach(), 'reg_local': reg.detach()}
        return bpr + reg_weight * reg + cl_weight * cl

    def predict_topk(self, users, k, trn_csr=None, topk_fn=None):
        """all-rank evaluation (full_predict + _mask_predict + t.topk, lightgcn.py:58-66, trainer/metrics.py:99-103):
        the tables are assembled once (`full_tables`) and every rank runs the fused top-k kernel on its share of the
        users; returns (ids [B_r, k], users of this rank) -- evaluation is embarrassingly parallel over users"""
        ue, ie = self.full_tables()
        mine = users[self.rank::self.world]
        return (topk_fn or ops.eval_topk)(ue.contiguous(), ie.contiguous(), mine, k, trn_csr), mine


class _GatherColumnsMultiFn(torch.autograd.Function):
    """_GatherColumnsFn for several (table slice, ids) pairs at once: ONE all-gather for all of them.
    args = (slice_0, ids_0, slice_1, ids_1, ...); returns one full-width [K_i, P*w] tensor per pair."""

    @staticmethod
    def forward(ctx, world, rank, group, scatter_fn, *args):
        tables, ids = args[0::2], args[1::2]
        parts = [t.index_select(0, i) for t, i in zip(tables, ids)]
        counts = [p.shape[0] for p in parts]
        mine = torch.cat(parts)
        ctx.save_for_backward(*ids)
        ctx.meta = (world, rank, scatter_fn, [t.shape[0] for t in tables], mine.shape[1])
        if not solo(world):
            K, w = mine.shape
            mine = all_gather_rows(mine, world, group).view(world, K, w).permute(1, 0, 2).reshape(K, world * w)
        return tuple(mine.split(counts))

    @staticmethod
    def backward(ctx, *grads):
        ids = ctx.saved_tensors
        world, rank, scatter_fn, n_rows, w = ctx.meta
        out = [None, None, None, None]
        for g, i, n in zip(grads, ids, n_rows):
            out += [scatter_fn(g[:, rank * w:(rank + 1) * w].contiguous(), i, n), None]
        return tuple(out)


class FeatureSlicedLightGCL(torch.nn.Module):
    """LightGCL (reference models/general_cf/lightgcl.py:73-125) on feature-sliced tables: ALL of its propagation is
    column-independent -- the two products A E_i and A^T E_u of a layer (lightgcl.py:58-65, 79-82) and the rank-q SVD view
    `u_mul_s @ (vt @ E)` (lightgcl.py:83-84, left multiplications) -- so a rank propagates its d/P columns of both tables and
    of both views with NO collective (the row-sharded form, shard.ShardedLightGCL, needs two table exchanges and two q x d
    all-reduces per layer and direction).  Losses (lightgcl.py:99-125): the batch rows of the four tables come from ONE small
    all-gather; the un-normalized InfoNCE needs all d columns of every user / item row, so E_u and E_i are transposed once per
    step to row blocks (`to_row_shards`) and the staged variant-1 kernels take over (B row sums forward, B x d anchor gradients
    backward all-reduced).  `graph_ui`: the WHOLE U x I adjacency (sslrec_amd.graph.PropGraph; `.transposed()` = A^T);
    `factors` = (ut [q, U], vt [q, I], u_mul_s [U, q], v_mul_s [I, q]), replicated.  `spmm_fn(graph, x)`, `lowrank_fn(left,
    right, x)`, `scatter_fn`, and the loss functions are injectable (gloo tests); defaults are the HIP ops.
    Status: the partition / collective logic is verified under gloo against the oracle's LightGCL step
    (tests/test_shard_gloo.py); a run with the real kernels on a GPU is still to be done."""

    def __init__(self, graph_ui, init_users, init_items, factors, layer_num, temp, world, rank, group=None, device=None,
                 spmm_fn=None, lowrank_fn=None, scatter_fn=None):
        super().__init__()
        self.graph, self.graph_t = graph_ui, graph_ui.transposed()
        self.layer_num, self.temp = int(layer_num), float(temp)
        self.world, self.rank, self.group = int(world), int(rank), group
        self.d = int(init_users.shape[1])
        self.lo, self.hi = slice_bounds(self.d, self.world, self.rank)
        device = torch.device(device if device is not None else getattr(graph_ui, 'device', 'cpu'))
        self.local_user_embeds = torch.nn.Parameter(init_users[:, self.lo:self.hi].detach().to(device).contiguous())
        self.local_item_embeds = torch.nn.Parameter(init_items[:, self.lo:self.hi].detach().to(device).contiguous())
        self.ut, self.vt, self.u_mul_s, self.v_mul_s = (f.to(device).contiguous() for f in factors)
        self.spmm_fn = spmm_fn or ops.spmm
        self.lowrank_fn = lowrank_fn or ops.lowrank_apply
        self.scatter_fn = scatter_fn or (_default_scatter if device.type == 'cuda' else _host_scatter)
        self.last_parts = {}

    def forward(self):
        """this rank's columns of (E_u, E_i, G_u, G_i): the layer sums of the graph view and of the SVD view"""
        e_u, e_i = [self.local_user_embeds], [self.local_item_embeds]
        g_u, g_i = [self.local_user_embeds], [self.local_item_embeds]
        for _ in range(self.layer_num):
            z_u = self.spmm_fn(self.graph, e_i[-1])                     # A   @ E_i
            z_i = self.spmm_fn(self.graph_t, e_u[-1])                   # A^T @ E_u
            g_u.append(self.lowrank_fn(self.u_mul_s, self.vt, e_i[-1]))
            g_i.append(self.lowrank_fn(self.v_mul_s, self.ut, e_u[-1]))
            e_u.append(z_u)
            e_i.append(z_i)
        return sum(e_u), sum(e_i), sum(g_u), sum(g_i)

    def _reduce(self, t):
        if not solo(self.world):
            all_reduce_sum(t, self.group)
        return t

    def _infonce(self, e1, e2, s_slice, infonce_fn):
        all_local = to_row_shards(s_slice, self.world, self.rank, self.group)
        if infonce_fn is not None:
            return infonce_fn(e1, e2, all_local, self.temp)
        return ops.infonce_loss_sharded(e1, e2, all_local, self.temp, 1, self._reduce)

    def lightgcl_loss(self, batch, cl_weight, reg_weight, extra_params=(), bpr_fn=None, reg_fn=None, infonce_fn=None):
        """bpr + cl_weight * cl + reg_weight * (this rank's share of the regularizer; the replicated `extra_params` -- the
        reference's unused-but-regularized Ws -- count once, on rank 0)"""
        ancs, poss, negs = batch[:3]
        B = ancs.shape[0]
        e_u, e_i, g_u, g_i = self.forward()
        anc, gu_a, pn, gi_p = _GatherColumnsMultiFn.apply(self.world, self.rank, self.group, self.scatter_fn,
                                                          e_u, ancs, g_u, ancs, e_i, torch.cat([poss, negs]), g_i, poss)
        bpr = (bpr_fn or (lambda a, p, n: ops.bpr_loss(a, p, n, variant=1)))(anc, pn[:B], pn[B:]) / B
        cl = (self._infonce(gu_a, anc, e_u, infonce_fn) + self._infonce(gi_p, pn[:B], e_i, infonce_fn)) / B
        sq = reg_fn or ops.sum_squares
        reg = sq(self.local_user_embeds) + sq(self.local_item_embeds)
        if self.rank == 0:
            for w in extra_params:
                reg = reg + sq(w)
        self.last_parts = {'bpr_loss': bpr.detach(), 'cl_loss': (cl_weight * cl).detach(), 'reg_local': reg.detach()}
        return bpr + cl_weight * cl + reg_weight * reg


class GraphedLightGCNStep:
    """LightGCN's cal_loss + backward (lightgcn.py:45-56) on feature-sliced tables as TWO captured hipGraphs around the step's
    one collective -- the ~40 eager launches of the step cost more host time than a GPU's shrinking share of the work takes:

        graph A:  L fused products of the slice (no autograd) -> the batch's [3B, d/P] rows
        eager  :  all-gather of those rows ([3B, d] on every rank)
        graph B:  BPR forward + backward on the dense rows (ops.bpr_loss_and_grads), this rank's gradient columns scattered
                  into [N, d/P], the backward recurrence g <- G + A^T g (L fused products), + 2 * reg_weight * E0

    Same kernels and the same arithmetic as `FeatureSlicedGraphCF.lightgcn_loss(...).backward()` (the regularizer's gradient is
    added with one axpy instead of a kernel of its own: equal to rounding).  After `step(batch)`: `model.local_embeds.grad` holds
    the gradient, `loss_bpr` / `reg_local` the loss parts (device scalars; the total is bpr + reg_weight * all-reduced reg).
    The batch size is fixed at construction; the graphs hold the addresses of the parameter, so an optimizer must update it in
    place (torch.optim and sslrec_amd.optim do)."""

    def __init__(self, model, batch_size, reg_weight, stamps=None):
        """stamps: an ops.StampLog -- the SpMM launches captured into the two graphs then time themselves with the device's wall
        clock on every replay (bench.py's per-launch roofline inside the timed region)"""
        self.model, self.B, self.reg_weight = model, int(batch_size), float(reg_weight)
        m = model
        e0 = m.local_embeds
        dev = e0.device
        if dev.type != 'cuda':
            raise RuntimeError('hipGraph capture needs a HIP device')
        n, w = e0.shape
        K = 3 * self.B
        self.ids = torch.zeros(K, dtype=torch.int64, device=dev)
        self.rows_local = torch.zeros((K, w), dtype=torch.float32, device=dev)
        self.rows_all = self.rows_local if solo(m.world) else torch.zeros((m.world * K, w), dtype=torch.float32, device=dev)
        self.grad = torch.zeros((n, w), dtype=torch.float32, device=dev)
        self.loss_bpr = torch.zeros(1, dtype=torch.float32, device=dev)
        self.reg_local = torch.zeros((), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):               # warm-up: layouts built, kernel attributes set, allocator primed
            for _ in range(2):
                self._part_a()
                self._part_b()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread may query its own events while this thread captures
        ops.STAMPS = stamps
        try:
            with torch.cuda.graph(self.graph_a, capture_error_mode='thread_local'):
                self._part_a()
            with torch.cuda.graph(self.graph_b, capture_error_mode='thread_local'):
                self._part_b()
        finally:
            ops.STAMPS = None

    def _part_a(self):
        m = self.model
        with torch.no_grad():
            s = m.propagate_fn(m.graph, m.local_embeds, m.layer_num)
            torch.index_select(s, 0, self.ids, out=self.rows_local)

    def _part_b(self):
        m, B = self.model, self.B
        with torch.no_grad():
            K, w = self.rows_local.shape
            full = self.rows_all if solo(m.world) else self.rows_all.view(m.world, K, w).permute(1, 0, 2).reshape(K, m.world * w)
            loss, da, dp, dn = ops.bpr_loss_and_grads(full[:B], full[B:2 * B], full[2 * B:], divisor=B)
            g_rows = torch.cat([da, dp, dn])[:, m.lo:m.hi].contiguous()
            G = m.scatter_fn(g_rows, self.ids, self.grad.shape[0])
            g = G
            for _ in range(m.layer_num):            # g_{l-1} = G + A^T g_l  (ops._PropagateSumFn.backward)
                nxt = torch.empty_like(G)
                ops.spmm_raw(m.graph, g, 'bwd', acc_in=G, acc_out=nxt, want_y=False)
                g = nxt
            e0 = m.local_embeds.detach()
            self.reg_local.copy_(ops.sum_squares(e0))
            torch.add(g, e0, alpha=2.0 * self.reg_weight, out=self.grad)
            self.loss_bpr.copy_(loss)

    def step(self, batch):
        m, B = self.model, self.B
        ancs, poss, negs = batch[:3]
        if ancs.shape[0] != B:
            raise ValueError('this step was captured for batches of %d, got %d' % (B, ancs.shape[0]))
        self.ids[:B].copy_(ancs)
        torch.add(poss, m.n_user, out=self.ids[B:2 * B])
        torch.add(negs, m.n_user, out=self.ids[2 * B:])
        self.graph_a.replay()
        if not solo(m.world):
            if dist.get_backend(m.group) == 'gloo':          # tests: host-staged
                self.rows_all.copy_(all_gather_rows(self.rows_local, m.world, m.group))
            else:
                dist.all_gather_into_tensor(self.rows_all, self.rows_local, group=m.group)
        self.graph_b.replay()
        m.local_embeds.grad = self.grad
        return self.loss_bpr
