"""world_size-2 test of the row-sharded propagation (sslrec_amd/shard.py) on CPU with the gloo
backend.  The local SpMM is injected (an oracle-side CSR product over the shard's own plan
arrays), so what is under test is the partition, the column re-labelling, the all-gather
layout and the backward recurrence -- the pieces that cannot be seen on a single GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_plan_spmm(plan_graph, xg, acc_in, acc_out, want_y, noise=None, eps=0.0):
    """oracle-side stand-in for ops.spmm_raw: walks the SAME packed layout on the host (fp32,
    per-lane-group partial sums added in a fixed order); optional EmbedPerturb epilogue"""
    sys.path.insert(0, ROOT)
    from tests.helpers import walk_packed
    y = torch.from_numpy(walk_packed(plan_graph.fwd.packed(32), xg.numpy(), dtype=np.float32))
    if noise is not None:
        from oracle import ref_expr as R
        y = R.embed_perturb(y, eps, noise)
    if acc_out is not None:
        acc_out.copy_(acc_in + y)
    return y if want_y else None


class _CpuShardedInfoNce(torch.autograd.Function):
    """test-side stand-in for ops.infonce_loss_sharded (same staging: all-reduce of the B row sums
    forward, of the B x d anchor-gradient partials backward), written with torch autograd"""

    @staticmethod
    def forward(ctx, e1, e2, all_local, temp):
        with torch.enable_grad():
            a, b, c = (t.detach().clone().requires_grad_(True) for t in (e1, e2, all_local))
            nrm = lambda x: x / torch.sqrt(1e-8 + x.square().sum(-1, keepdim=True))
            an, bn, cn = nrm(a), nrm(b), nrm(c)
            pos = (an * bn).sum(-1) / temp
            z_loc = torch.exp(an @ cn.T / temp).sum(-1)
        z = z_loc.detach().clone()
        dist.all_reduce(z)
        ctx.graph = (a, b, c, pos, z_loc, z)
        return (-pos.detach() + torch.log(z)).sum()

    @staticmethod
    def backward(ctx, g):
        a, b, c, pos, z_loc, z = ctx.graph
        with torch.enable_grad():
            ga_z, gc = torch.autograd.grad((z_loc / z).sum(), [a, c], retain_graph=True)
            ga_p, gb = torch.autograd.grad(-pos.sum(), [a, b])
        ga_z = ga_z.clone()
        dist.all_reduce(ga_z)
        return g * (ga_z + ga_p), g * gb, g * gc, None


def _cpu_topk(ue, ie, users, k, csr, return_scores=False):
    """test-side stand-in for ops.eval_topk (sslrec_eval_topk_f32): scores of ALL rows of `ue` against the item rows
    `ie`, train items (csr = (rowptr, col) per row of ue) excluded, k best by (score desc, item id asc), -1 / -inf where
    fewer than k remain"""
    sc = ue.double() @ ie.double().T
    if csr is not None:
        rowptr, col = csr
        for b in range(sc.shape[0]):
            sc[b, col[rowptr[b]:rowptr[b + 1]]] = float('-inf')
    kk = min(k, sc.shape[1])
    order = (-sc).argsort(dim=1, stable=True)[:, :kk]
    val = sc.gather(1, order)
    idx = torch.where(torch.isinf(val), torch.full_like(order, -1), order)
    if kk < k:
        idx = torch.cat([idx, torch.full((sc.shape[0], k - kk), -1, dtype=idx.dtype)], 1)
        val = torch.cat([val, torch.full((sc.shape[0], k - kk), float('-inf'), dtype=val.dtype)], 1)
    return (idx, val.float()) if return_scores else idx


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import ref_expr as R
        from sslrec_amd.data_utils.synth import make_dataset
        from sslrec_amd.graph import PropGraph
        from sslrec_amd.shard import ShardedGraph, local_rows, sharded_propagate_sum
        trn = R.binarize_coo(make_dataset('tiny', seed=3))
        idx, vals, n = R.normalized_bipartite_coo(trn)
        # make it ASYMMETRIC (like an edge-dropped view) so that A^T shards are really exercised
        keep = np.random.default_rng(0).random(vals.size) < 0.6
        rows, cols, v = idx[0][keep], idx[1][keep], vals[keep]
        L, d = 3, 32
        gen = torch.Generator().manual_seed(1)
        e0 = torch.randn(n, d, generator=gen)
        w = torch.randn(n, d, generator=gen)
        # single-process result over the same plan machinery (seg_max small -> long rows too)
        full = PropGraph._single(rows, cols, v, (n, n), 'cpu', seg_max=8)
        full_t = PropGraph._single(cols, rows, v, (n, n), 'cpu', seg_max=8)
        x, tot = e0, e0.clone()
        for _ in range(L):
            x = _cpu_plan_spmm(full, x, None, None, True)
            tot = tot + x
        g = w.clone()
        for _ in range(L):
            g = w + _cpu_plan_spmm(full_t, g, None, None, True)
        # sharded
        sg = ShardedGraph(rows, cols, v, n, world, rank, 'cpu', seg_max=8)
        e0_loc = sg.to_local(e0).requires_grad_(True)
        tot_loc = sharded_propagate_sum(sg, e0_loc, L, spmm_fn=_cpu_plan_spmm)
        (tot_loc * sg.to_local(w)).sum().backward()
        ids = local_rows(n, world, rank)
        ok_f = torch.equal(tot_loc.detach()[:ids.size], tot[ids])          # bit-identical to 1 process
        ok_b = torch.equal(e0_loc.grad[:ids.size], g[ids])
        # shard-local construction (ShardedGraph.from_local_entries): a rank brings only the pattern entries of its own rows of
        # A and of A^T, the degrees are all-gathered once -- same shard matrices, values bit-identical to the data handler's,
        # same propagation; the three formulations work from the local entry lists
        pr, pc = idx[0], idx[1]                                      # the whole normalized adjacency (symmetric pattern)
        mine_f, mine_b = pr % world == rank, pc % world == rank
        sl = ShardedGraph.from_local_entries((pr[mine_f], pc[mine_f]), (pr[mine_b], pc[mine_b]), n, world, rank, 'cpu', seg_max=8)
        sgw = ShardedGraph(pr, pc, vals, n, world, rank, 'cpu', seg_max=8)
        for a_, b_ in ((sl.a, sgw.a), (sl.at, sgw.at)):
            ok_f = ok_f and np.array_equal(a_.fwd.rowptr_host, b_.fwd.rowptr_host) and np.array_equal(a_.fwd.csr_col_host, b_.fwd.csr_col_host)
            ok_f = ok_f and np.array_equal(a_.fwd.csr_val_host.view(np.int32), b_.fwd.csr_val_host.view(np.int32))
        ok_f = ok_f and np.array_equal(np.sort(sl.coo_ids_fwd), np.sort(sl.coo_ids_bwd)) if world == 1 else ok_f
        for mode in ('all_gather', 'pipelined', 'reduce_scatter'):
            xa = sl.to_local(e0).requires_grad_(True)
            ta = sharded_propagate_sum(sl, xa, L, spmm_fn=_cpu_plan_spmm, mode=mode)
            (ta * sl.to_local(w)).sum().backward()
            xb = sgw.to_local(e0).requires_grad_(True)
            tb = sharded_propagate_sum(sgw, xb, L, spmm_fn=_cpu_plan_spmm, mode=mode)
            (tb * sgw.to_local(w)).sum().backward()
            ok_f = ok_f and torch.equal(ta.detach(), tb.detach())
            ok_b = ok_b and torch.equal(xa.grad, xb.grad)
        # the EdgeDrop ids of an entry agree between the A shard of its row's owner and the A^T shard of its column's owner
        from sslrec_amd.shard import entry_key
        keys = entry_key(pr, pc)
        ok_f = ok_f and np.array_equal(sl.coo_ids_fwd, keys[mine_f]) and np.array_equal(sl.coo_ids_bwd, keys[mine_b])
        ok_f = ok_f and int(keys.max()) < 2 ** 31 and np.unique(keys).size > 0.999 * keys.size
        # reduce-scatter dual (column-sharded A): same result up to the order of the P-way fp32 sum
        e0_rs = sg.to_local(e0).requires_grad_(True)
        tot_rs = sharded_propagate_sum(sg, e0_rs, L, spmm_fn=_cpu_plan_spmm, mode='reduce_scatter')
        (tot_rs * sg.to_local(w)).sum().backward()
        ok_f = ok_f and torch.allclose(tot_rs.detach()[:ids.size], tot[ids], rtol=0, atol=1e-5)
        ok_b = ok_b and torch.allclose(e0_rs.grad[:ids.size], g[ids], rtol=0, atol=1e-5)
        ok_f = ok_f and bool((tot_rs.detach()[ids.size:] == 0).all())       # padding rows stay zero
        # pipelined exchange (one broadcast per source rank, block products as the shards arrive): equal to rounding
        e0_pp = sg.to_local(e0).requires_grad_(True)
        tot_pp = sharded_propagate_sum(sg, e0_pp, L, spmm_fn=_cpu_plan_spmm, mode='pipelined')
        (tot_pp * sg.to_local(w)).sum().backward()
        ok_f = ok_f and torch.allclose(tot_pp.detach()[:ids.size], tot[ids], rtol=0, atol=1e-5)
        ok_b = ok_b and torch.allclose(e0_pp.grad[:ids.size], g[ids], rtol=0, atol=1e-5)
        # full sharded LightGCN step (batch rows exchanged by one small all-reduce) vs the oracle step
        from sslrec_amd.shard import ShardedGraphCF
        n_user = trn.shape[0]
        sgs = ShardedGraph(idx[0], idx[1], vals, n, world, rank, 'cpu', seg_max=8)
        model = ShardedGraphCF(sgs, n_user, n - n_user, e0, 2, spmm_fn=_cpu_plan_spmm)
        B = 37
        batch = [torch.randint(0, n_user, (B,), generator=gen), torch.randint(0, n - n_user, (B,), generator=gen),
                 torch.randint(0, n - n_user, (B,), generator=gen)]
        sq = lambda w: w.square().sum()
        loss = model.lightgcn_loss(batch, 1e-3, bpr_fn=R.cal_bpr_loss, reg_fn=sq)
        loss.backward()
        reg = model.last_parts['reg_local'].clone()
        dist.all_reduce(reg)
        total = model.last_parts['bpr_loss'] + 1e-3 * reg
        adj = R.torch_adj_from(idx, vals, n)
        ue = e0[:n_user].clone().requires_grad_(True); ie = e0[n_user:].clone().requires_grad_(True)
        ref_loss, _ = R.lightgcn_cal_loss(adj, ue, ie, batch, 2, 1.0, 1e-3)
        ref_loss.backward()
        ref_grad = torch.cat([ue.grad, ie.grad])
        ok_f = ok_f and abs(total.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
        ok_b = ok_b and torch.allclose(model.local_embeds.grad[:ids.size], ref_grad[ids], rtol=1e-4, atol=1e-6)
        ok_b = ok_b and bool((model.local_embeds.grad[ids.size:] == 0).all())
        # the all-gathered tables used for evaluation: global row order restored on every rank
        with torch.no_grad():
            users, items = model.tables()
            ru, ri = R.lightgcn_forward(adj, e0[:n_user], e0[n_user:], 2)
        ok_f = ok_f and torch.allclose(users, ru, atol=1e-5) and torch.allclose(items, ri, atol=1e-5)
        # evaluation with the item table kept sharded: per-rank top-k over the rank's items + merge == top-k over everything
        n_item, k_eval = n - n_user, 20
        eval_users = torch.randint(0, n_user, (41,), generator=gen)
        trn_csr = trn.tocsr()
        got_ids, got_val = model.predict_topk(eval_users, k_eval, model.local_train_csr(trn_csr), topk_fn=_cpu_topk)
        seen = torch.from_numpy(trn_csr[eval_users.numpy()].toarray() != 0)
        full = (ru[eval_users].double() @ ri.double().T).masked_fill(seen, float('-inf'))
        want_val, want_ids = torch.topk(full, k_eval)
        ok_f = ok_f and torch.allclose(got_val.double(), want_val, rtol=1e-5, atol=1e-5)
        ok_f = ok_f and not bool(seen.gather(1, got_ids.clamp(min=0))[got_ids >= 0].any())      # never a train item
        ok_f = ok_f and bool((got_ids == want_ids)[(want_val[:, :-1] - want_val[:, 1:]).min(1).values > 1e-6].all())
        ok_f = ok_f and sorted(model.local_item_ids().tolist() + [i for i in range(n_item) if (i + n_user) % world != rank]) == list(range(n_item))
        # sharded SimGCL step: perturbed views, exchanged batch rows, InfoNCE with `all` kept sharded
        model.local_embeds.grad = None
        nz = [[torch.rand(n, d, generator=gen) for _ in range(2)] for _ in range(2)]
        loc = [[sgs.to_local(t) for t in view] for view in nz]
        import sslrec_amd.shard as shard_mod
        gathers, plain_gather = [0], shard_mod.all_gather_rows

        def counted_gather(*a, **k):
            gathers[0] += 1
            return plain_gather(*a, **k)
        shard_mod.all_gather_rows = counted_gather
        try:
            loss = model.simgcl_loss(batch, loc[0], loc[1], 0.1, 1e-3, 0.2, 0.5, bpr_fn=R.cal_bpr_loss, reg_fn=sq,
                                     infonce_fn=_CpuShardedInfoNce.apply)
            loss.backward()
        finally:
            shard_mod.all_gather_rows = plain_gather
        # L = 2, three views: E0 gathered once + one more exchange per view going forward (4 instead of 6), ONE backward chain of 2
        # exchanges for the three views (instead of 6)
        ok_f = ok_f and gathers[0] == 4 + 2
        ue = e0[:n_user].clone().requires_grad_(True); ie = e0[n_user:].clone().requires_grad_(True)
        ref_loss, ref_parts = R.simgcl_cal_loss(adj, ue, ie, batch, 2, 1e-3, 0.2, 0.5, 0.1, noise_draws=(nz[0], nz[1]))
        ref_loss.backward()
        ref_grad = torch.cat([ue.grad, ie.grad])
        ok_f = ok_f and abs(0.2 * model.last_parts['cl_loss'].item() - ref_parts['cl_loss'].item()) <= 1e-5 * abs(ref_parts['cl_loss'].item())
        ok_f = ok_f and abs(model.last_parts['bpr_loss'].item() - ref_parts['bpr_loss'].item()) <= 1e-5 * abs(ref_parts['bpr_loss'].item())
        ok_b = ok_b and torch.allclose(model.local_embeds.grad[:ids.size], ref_grad[ids], rtol=1e-4, atol=1e-6)
        counts = torch.tensor([sg.nnz_local], dtype=torch.int64)
        dist.all_reduce(counts)
        q.put((rank, bool(ok_f), bool(ok_b), int(counts.item()), int(rows.size)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_propagation_two_ranks_bitwise_equal_to_one(world):
    """(world 8 = the node this is built for: the 8-way cyclic deal, every collective with 8 contributions)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_f, ok_b, total_nnz, nnz in res:
        assert ok_f, 'forward shard of rank %d differs from the single-process result' % rank
        assert ok_b, 'backward shard of rank %d differs from the single-process result' % rank
        assert total_nnz == nnz          # every entry is owned by exactly one rank


def test_partition_helpers():
    from sslrec_amd.shard import gathered_position, local_rows, rows_per_rank
    n, world = 23, 4
    owned = np.concatenate([local_rows(n, world, r) for r in range(world)])
    assert sorted(owned.tolist()) == list(range(n))
    pos = gathered_position(np.arange(n), n, world)
    assert len(set(pos.tolist())) == n and pos.max() < rows_per_rank(n, world) * world
    for r in range(world):
        ids = local_rows(n, world, r)
        assert np.array_equal(gathered_position(ids, n, world), r * rows_per_rank(n, world) + np.arange(ids.size))


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE config 5 at small scale: LightGCL on row-sharded tables, built shard-locally
# ----------------------------------------------------------------------------------------------------------------------
class _CpuShardedInfoNceV1(torch.autograd.Function):
    """test-side stand-in for ops.infonce_loss_sharded(variant=1) (LightGCL's un-normalized form, lightgcl.py:114-118),
    same staging: the B row sums all-reduced forward, the B x d anchor-gradient partials backward"""

    @staticmethod
    def forward(ctx, e1, e2, all_local, temp):
        with torch.enable_grad():
            a, b, c = (t.detach().clone().requires_grad_(True) for t in (e1, e2, all_local))
            pos = torch.clamp((a * b).sum(-1) / temp, -5.0, 5.0)
            z_loc = torch.exp(a @ c.T / temp).sum(-1)
        z = z_loc.detach().clone()
        dist.all_reduce(z)
        ctx.graph = (a, b, c, pos, z_loc, z)
        return (-pos.detach() + torch.log(z + 1e-8)).sum()

    @staticmethod
    def backward(ctx, g):
        a, b, c, pos, z_loc, z = ctx.graph
        with torch.enable_grad():
            ga_z, gc = torch.autograd.grad((z_loc / (z + 1e-8)).sum(), [a, c], retain_graph=True)
            ga_p, gb = torch.autograd.grad(-pos.sum(), [a, b])
        ga_z = ga_z.clone()
        dist.all_reduce(ga_z)
        return g * (ga_z + ga_p), g * gb, g * gc, None


class _CpuRankQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, left_local, right_local, reduce):
        s_ = right_local @ x_local
        reduce(s_)
        ctx.save_for_backward(left_local, right_local)
        ctx.reduce = reduce
        return left_local @ s_

    @staticmethod
    def backward(ctx, gy):
        left_local, right_local = ctx.saved_tensors
        s_ = left_local.T @ gy
        ctx.reduce(s_)
        return right_local.T @ s_, None, None, None


def _cpu_rankq(left_local, right_local, x_local, reduce):
    return _CpuRankQ.apply(x_local, left_local, right_local, reduce)


def _lightgcl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        from oracle import ref_expr as R
        from sslrec_amd.data_utils.synth import cell_bipartite, sharded_cells
        from sslrec_amd.shard import ShardedBipartite, ShardedLightGCL, local_rows
        U, I, E, d, L, q_rank, temp = 1203, 1571, 24000, 32, 2, 5, 0.5      # ~1/8000 of cfg 5's 10 M x 10 M
        fwd, bwd = sharded_cells(U, I, E, world, rank, seed=11)
        sb = ShardedBipartite.from_local_entries(fwd, bwd, U, I, world, rank, 'cpu', seg_max=8)
        # the same graph on one process: the union of all cells
        cells = [cell_bipartite(U, I, E, world, a, b, seed=11) for a in range(world) for b in range(world)]
        gu, gi = np.concatenate([c[0] for c in cells]), np.concatenate([c[1] for c in cells])
        assert len(set((gu * I + gi).tolist())) == gu.size                   # cells are disjoint
        trn = sp.coo_matrix((np.ones(gu.size, dtype=np.float32), (gu, gi)), shape=(U, I))
        adj = R.lightgcl_adj(trn)
        gen = torch.Generator().manual_seed(2)
        ue, ie = torch.randn(U, d, generator=gen) * 0.1, torch.randn(I, d, generator=gen) * 0.1
        ws = [torch.randn(d, d, generator=gen) * 0.1 for _ in range(L)]
        ut, vt = torch.randn(q_rank, U, generator=gen) * 0.05, torch.randn(q_rank, I, generator=gen) * 0.05
        u_mul_s, v_mul_s = torch.randn(U, q_rank, generator=gen) * 0.05, torch.randn(I, q_rank, generator=gen) * 0.05
        factors = (sb.local_users(ut.T.contiguous()).T.contiguous(), sb.local_items(vt.T.contiguous()).T.contiguous(),
                   sb.local_users(u_mul_s), sb.local_items(v_mul_s))
        model = ShardedLightGCL(sb, ue, ie, factors, L, temp, spmm_fn=_cpu_plan_spmm, rankq_fn=_cpu_rankq, add_fn=lambda out, a, b: out.copy_(a + b))
        B = 61
        batch = [torch.randint(0, U, (B,), generator=gen), torch.randint(0, I, (B,), generator=gen),
                 torch.randint(0, I, (B,), generator=gen)]
        w_params = [w.clone().requires_grad_(True) for w in ws]
        sq = lambda w: w.square().sum()
        loss = model.lightgcl_loss(batch, 0.2, 1e-3, extra_params=w_params, bpr_fn=lambda a, p, n: R.lightgcl_bpr(a, p, n) * B,
                                   reg_fn=sq, infonce_fn=_CpuShardedInfoNceV1.apply)
        loss.backward()
        reg = model.last_parts['reg_local'].clone()
        dist.all_reduce(reg)
        total = model.last_parts['bpr_loss'] + model.last_parts['cl_loss'] + 1e-3 * reg
        # oracle: the reference's LightGCL step on the whole graph (lightgcl.py:73-125)
        rue, rie = ue.clone().requires_grad_(True), ie.clone().requires_grad_(True)
        rws = [w.clone().requires_grad_(True) for w in ws]
        ref_loss, ref_parts = R.lightgcl_cal_loss(adj, rue, rie, rws, (ut, vt, u_mul_s, v_mul_s), batch, L, 1e-3, 0.2, temp)
        ref_loss.backward()
        uid, iid = local_rows(U, world, rank), local_rows(I, world, rank)
        ok_f = abs(total.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item())
        ok_f = ok_f and abs(model.last_parts['cl_loss'].item() - ref_parts['cl_loss'].item()) <= 2e-5 * abs(ref_parts['cl_loss'].item())
        ok_b = torch.allclose(model.local_user_embeds.grad[:uid.size], rue.grad[uid], rtol=1e-4, atol=1e-7)
        ok_b = ok_b and torch.allclose(model.local_item_embeds.grad[:iid.size], rie.grad[iid], rtol=1e-4, atol=1e-7)
        ok_b = ok_b and bool((model.local_user_embeds.grad[uid.size:] == 0).all())
        # the graph view is BIT-IDENTICAL to one process walking the same layouts (no reduction crosses a rank)
        with torch.no_grad():
            e_u, e_i, g_u, g_i = model.forward()
            one = ShardedBipartite(gu, gi, _lightgcl_vals(gu, gi, U, I), U, I, 1, 0, 'cpu', seg_max=8)
            ref_m = ShardedLightGCL(one, ue, ie, (ut, vt, u_mul_s, v_mul_s), L, temp, spmm_fn=_cpu_plan_spmm, rankq_fn=_cpu_rankq, add_fn=lambda out, a, b: out.copy_(a + b))
            r_u, r_i, rg_u, rg_i = ref_m.forward()
        ok_f = ok_f and torch.equal(e_u[:uid.size], r_u[uid]) and torch.equal(e_i[:iid.size], r_i[iid])
        ok_f = ok_f and torch.allclose(g_u[:uid.size], rg_u[uid], rtol=0, atol=1e-6)
        # round 6: the graph view as ONE node (layer sums in the products' epilogues, hand-written backward): the same bits as the
        # separate product nodes of rounds 4-5; with the PIPELINED exchange (per-source-rank broadcasts + block products, sums grouped by
        # source rank) the same numbers to rounding
        cpu_add = lambda out, a, b: out.copy_(a + b)
        grads = {}
        cpu_lowrank = (lambda m, tr, x: (m.T if tr else m) @ x, lambda m, tr, s_: (m if tr else m.T) @ s_)
        for mode in ('separate', 'all_gather', 'pipelined', 'all_gather+svd_node'):
            # ('+svd_node': the SVD view of a table as ONE node -- the L partial q x d products added before one all-reduce and one expansion)
            m2 = ShardedLightGCL(sb, ue, ie, factors, L, temp, spmm_fn=_cpu_plan_spmm, rankq_fn=_cpu_rankq, mode=mode.split('+')[0], add_fn=cpu_add,
                                 lowrank_ops=cpu_lowrank if '+' in mode else None)
            w2 = [w.clone().requires_grad_(True) for w in ws]
            l2 = m2.lightgcl_loss(batch, 0.2, 1e-3, extra_params=w2, bpr_fn=lambda a, p, n: R.lightgcl_bpr(a, p, n) * B,
                                  reg_fn=sq, infonce_fn=_CpuShardedInfoNceV1.apply)
            l2.backward()
            with torch.no_grad():
                tabs = m2.forward()
            grads[mode] = (l2.detach(), m2.local_user_embeds.grad.clone(), m2.local_item_embeds.grad.clone(), [t.clone() for t in tabs])
        sep, fused, pipe = grads['separate'], grads['all_gather'], grads['pipelined']
        ok_f = ok_f and all(torch.equal(a, b) for a, b in zip(sep[3][:2], fused[3][:2]))       # graph-view sums: bit for bit
        ok_f = ok_f and all(torch.allclose(a, b, rtol=0, atol=1e-6) for a, b in zip(sep[3], fused[3]))
        ok_f = ok_f and all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(sep[3], pipe[3]))
        ok_f = ok_f and abs(sep[0].item() - fused[0].item()) <= 1e-6 * abs(sep[0].item()) and abs(sep[0].item() - pipe[0].item()) <= 2e-6 * abs(sep[0].item())
        svd = grads['all_gather+svd_node']
        ok_f = ok_f and all(torch.equal(a, b) for a, b in zip(sep[3][:2], svd[3][:2])) and all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(sep[3], svd[3]))
        ok_f = ok_f and abs(sep[0].item() - svd[0].item()) <= 2e-6 * abs(sep[0].item())
        for other in (fused, pipe, svd):
            ok_b = ok_b and torch.allclose(other[1], sep[1], rtol=1e-4, atol=1e-7) and torch.allclose(other[2], sep[2], rtol=1e-4, atol=1e-7)
            ok_b = ok_b and torch.allclose(other[1][:uid.size], rue.grad[uid], rtol=1e-4, atol=1e-7)
            ok_b = ok_b and torch.allclose(other[2][:iid.size], rie.grad[iid], rtol=1e-4, atol=1e-7)
        q.put((rank, bool(ok_f), bool(ok_b), float(total.item()), float(ref_loss.item())))
    finally:
        dist.destroy_process_group()


def _lightgcl_vals(users, items, n_user, n_item):
    du = np.bincount(users, minlength=n_user).astype(np.float32)
    di = np.bincount(items, minlength=n_item).astype(np.float32)
    return (1.0 / np.sqrt(du[users] * di[items])).astype(np.float32)


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_lightgcl_two_ranks_matches_the_oracle_step_and_one_rank(world):
    """config 5's path at ~1/8000 scale: shard-local generation (cells), degree exchange, sharded A / A^T products,
    rank-q view with its q x d all-reduce, sharded un-normalized InfoNCE -- loss and gradients vs the oracle's LightGCL
    step on the whole graph; the graph-view tables bitwise equal to the single-rank walk"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_lightgcl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_f, ok_b, total, ref in res:
        assert ok_f, 'rank %d: forward differs (loss %r vs oracle %r)' % (rank, total, ref)
        assert ok_b, 'rank %d: gradients differ from the oracle' % rank


# ---- feature-sliced tables (sslrec_amd/feature_shard.py): every rank holds all rows, d / P columns ---------------------
def _cpu_propagate_sum(adj, e0, layer_num, noises=None, eps=0.0, noise_sumsq=None, noise_geom=None):
    """test-side stand-in for ops.propagate_sum: torch.spmm over the oracle's COO adjacency, differentiable; with noises: the
    perturbation of a COLUMN SLICE (the row norm comes in as noise_sumsq, the way the kernels' epilogue takes it)"""
    x, tot = e0, e0
    for l in range(layer_num):
        x = torch.spmm(adj, x)
        if noises is not None:
            nrm = noise_sumsq[l].sqrt().clamp_min(1e-12)[:, None]
            x = x + (noises[l] / nrm) * torch.sign(x) * eps
        tot = tot + x
    return tot


def _feature_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import ref_expr as R
        from sslrec_amd.data_utils.synth import make_dataset
        from sslrec_amd.feature_shard import FeatureSlicedGraphCF, row_block, slice_bounds, to_row_shards
        trn = R.binarize_coo(make_dataset('tiny', seed=5))
        idx, vals, n = R.normalized_bipartite_coo(trn)
        n_user = trn.shape[0]
        n_item = n - n_user
        adj = R.torch_adj_from(idx, vals, n)
        L, d, B = 2, 32, 41
        gen = torch.Generator().manual_seed(7)
        e0 = torch.randn(n, d, generator=gen) * 0.1
        batch = [torch.randint(0, n_user, (B,), generator=gen), torch.randint(0, n_item, (B,), generator=gen),
                 torch.randint(0, n_item, (B,), generator=gen)]
        batch[0][:5] = batch[0][5]                                   # duplicated rows: the scatter of the backward adds them
        lo, hi = slice_bounds(d, world, rank)
        sq = lambda w: w.square().sum()
        model = FeatureSlicedGraphCF(adj, n_user, n_item, e0, L, world, rank, device='cpu', propagate_fn=_cpu_propagate_sum)
        ok = list(model.local_embeds.shape) == [n, d // world]
        # --- the transposition used by the contrastive terms: column slices -> row blocks and back
        t = torch.randn(n_item, d, generator=gen)
        mine = t[:, lo:hi].clone().requires_grad_(True)
        rows_blk = to_row_shards(mine, world, rank)
        a, b = row_block(n_item, world, rank)
        ok = ok and torch.equal(rows_blk.detach(), t[a:b])
        wgt = torch.randn(n_item, d, generator=gen)
        (rows_blk * wgt[a:b]).sum().backward()
        ok = ok and torch.equal(mine.grad, wgt[:, lo:hi])
        # --- LightGCN step vs the oracle step on the whole table
        loss = model.lightgcn_loss(batch, 1e-3, bpr_fn=R.cal_bpr_loss, reg_fn=sq)
        loss.backward()
        reg = model.last_parts['reg_local'].clone()
        dist.all_reduce(reg)
        total = model.last_parts['bpr_loss'] + 1e-3 * reg
        ue = e0[:n_user].clone().requires_grad_(True); ie = e0[n_user:].clone().requires_grad_(True)
        ref_loss, _ = R.lightgcn_cal_loss(adj, ue, ie, batch, L, 1.0, 1e-3)
        ref_loss.backward()
        ref_grad = torch.cat([ue.grad, ie.grad])
        ok_l = abs(total.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
        ok_g = torch.allclose(model.local_embeds.grad, ref_grad[:, lo:hi], rtol=1e-4, atol=1e-7)
        # --- tables for evaluation: global column order restored on every rank
        users, items = model.full_tables()
        ru, ri = R.lightgcn_forward(adj, e0[:n_user], e0[n_user:], L)
        ok = ok and torch.allclose(users, ru, atol=1e-6) and torch.allclose(items, ri, atol=1e-6)
        # --- SGL-ED step: two edge-dropped views (same recorded draws on every rank), InfoNCE through the transposition
        model.local_embeds.grad = None
        draws = [torch.rand(vals.size, generator=gen) for _ in range(2)]
        views = [R.edge_drop(adj, 0.7, dr) for dr in draws]
        loss = model.sgl_loss(batch, views[0], views[1], 1e-3, 0.3, 0.5, bpr_fn=R.cal_bpr_loss, reg_fn=sq,
                              infonce_fn=_CpuShardedInfoNce.apply)
        loss.backward()
        ue = e0[:n_user].clone().requires_grad_(True); ie = e0[n_user:].clone().requires_grad_(True)
        ref_loss, ref_parts = R.sgl_cal_loss(adj, ue, ie, batch, L, 0.7, 1e-3, 0.3, 0.5, mask_draws=draws)
        ref_loss.backward()
        ref_grad = torch.cat([ue.grad, ie.grad])
        ok_l = ok_l and abs(0.3 * model.last_parts['cl_loss'].item() - ref_parts['cl_loss'].item()) <= 1e-5 * abs(ref_parts['cl_loss'].item())
        ok_l = ok_l and abs(model.last_parts['bpr_loss'].item() - ref_parts['bpr_loss'].item()) <= 1e-5 * abs(ref_parts['bpr_loss'].item())
        ok_g = ok_g and torch.allclose(model.local_embeds.grad, ref_grad[:, lo:hi], rtol=1e-4, atol=1e-7)
        # --- SimGCL step: every rank holds ITS columns of the reference's two x L noise draws; the full-row norm of EmbedPerturb
        # (aug_utils.py:130) comes from an all-reduce of the ranks' partial sums of squares
        model.local_embeds.grad = None
        full_draws = [[torch.rand(n, d, generator=gen) for _ in range(L)] for _ in range(2)]
        mine = [[nz[:, lo:hi].contiguous() for nz in view] for view in full_draws]
        loss = model.simgcl_loss(batch, mine[0], mine[1], 0.2, 1e-3, 0.3, 0.5, bpr_fn=R.cal_bpr_loss, reg_fn=sq,
                                 infonce_fn=_CpuShardedInfoNce.apply, row_sumsq_fn=lambda t_: t_.square().sum(1))
        loss.backward()
        ue = e0[:n_user].clone().requires_grad_(True); ie = e0[n_user:].clone().requires_grad_(True)
        ref_loss, ref_parts = R.simgcl_cal_loss(adj, ue, ie, batch, L, 1e-3, 0.3, 0.5, 0.2, noise_draws=full_draws)
        ref_loss.backward()
        ref_grad = torch.cat([ue.grad, ie.grad])
        ok_l = ok_l and abs(0.3 * model.last_parts['cl_loss'].item() - ref_parts['cl_loss'].item()) <= 1e-5 * abs(ref_parts['cl_loss'].item())
        ok_l = ok_l and abs(model.last_parts['bpr_loss'].item() - ref_parts['bpr_loss'].item()) <= 1e-5 * abs(ref_parts['bpr_loss'].item())
        ok_g = ok_g and torch.allclose(model.local_embeds.grad, ref_grad[:, lo:hi], rtol=1e-4, atol=1e-7)
        # --- evaluation: tables assembled once, every rank ranks its share of the users (embarrassingly parallel over users)
        eval_users = torch.randint(0, n_user, (23,), generator=gen)
        trn_csr = trn.tocsr(); trn_csr.sort_indices()
        whole = (torch.from_numpy(trn_csr.indptr.astype(np.int64)), torch.from_numpy(trn_csr.indices.astype(np.int64)))
        got_ids, mine = model.predict_topk(eval_users, 10, whole, topk_fn=_cpu_topk_csr)
        ok = ok and torch.equal(mine, eval_users[rank::world])
        seen = torch.from_numpy(trn_csr[mine.numpy()].toarray() != 0)
        full = (ru[mine].double() @ ri.double().T).masked_fill(seen, float('-inf'))
        want_val, want_ids = torch.topk(full, 10)
        sep = (want_val[:, :-1] - want_val[:, 1:]).min(1).values > 1e-6
        ok = ok and bool((got_ids == want_ids)[sep].all()) and not bool(seen.gather(1, got_ids)[got_ids >= 0].any())
        q.put((rank, bool(ok), bool(ok_l), bool(ok_g)))
    finally:
        dist.destroy_process_group()


def _cpu_topk_csr(ue, ie, users, k, csr, return_scores=False):
    """test-side stand-in for ops.eval_topk with a whole-table train CSR (rowptr, col) indexed by GLOBAL user id"""
    rowptr, col = csr
    sc = ue[users].double() @ ie.double().T
    for r, u in enumerate(users.tolist()):
        sc[r, col[rowptr[u]:rowptr[u + 1]]] = float('-inf')
    val, idx = torch.topk(sc, k)
    return (idx, val.float()) if return_scores else idx


@pytest.mark.parametrize('world', [2, 4, 8])
def test_feature_sliced_steps_match_the_oracle(world):
    """FeatureSlicedGraphCF with gloo ranks: LightGCN, SGL-ED and SimGCL steps (loss parts, the rank's gradient columns)
    against the oracle's single-process steps; the slices -> row-blocks transposition and its backward are exact"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_feature_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ok_l, ok_g in res:
        assert ok, 'rank %d: slicing / transposition / assembled tables differ' % rank
        assert ok_l, 'rank %d: loss differs from the oracle step' % rank
        assert ok_g, 'rank %d: gradient columns differ from the oracle step' % rank


def _feature_traj_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import ref_expr as R
        from tests import helpers as H
        from sslrec_amd.feature_shard import FeatureSlicedGraphCF
        g, cfg, opt_cfg, meta = H.load_trajectory('lightgcn')
        dh = H.trajectory_setup('lightgcn', g, cfg, opt_cfg, meta, 'cpu')
        torch.set_num_threads(1)
        n_user, n_item = (int(x) for x in g['shape'])
        d = cfg['embedding_size']
        ue = torch.nn.init.xavier_uniform_(torch.empty(n_user, d))          # the reference's initialisation order
        ie = torch.nn.init.xavier_uniform_(torch.empty(n_item, d))
        model = FeatureSlicedGraphCF(dh.torch_adj, n_user, n_item, torch.cat([ue, ie]), cfg['layer_num'], world, rank,
                                     device='cpu', propagate_fn=_cpu_propagate_sum)
        opt = torch.optim.Adam([model.local_embeds], lr=opt_cfg['lr'], weight_decay=opt_cfg['weight_decay'])
        sq = lambda w: w.square().sum()
        losses = []
        for _ in range(meta['epochs']):
            dh.train_dataloader.dataset.sample_negs()
            for tem in dh.train_dataloader:
                batch = [x.long() for x in tem]
                opt.zero_grad()
                view = R.edge_drop(dh.torch_adj, cfg['keep_rate'])             # the reference's draw, same on every rank
                loss = model.lightgcn_loss(batch, cfg['reg_weight'], bpr_fn=R.cal_bpr_loss, reg_fn=sq, adj=view)
                loss.backward()
                opt.step()
                reg = model.last_parts['reg_local'].clone()
                dist.all_reduce(reg)
                losses.append(model.last_parts['bpr_loss'].item() + cfg['reg_weight'] * reg.item())
        users, items = model.full_tables(model.local_embeds.detach())
        ok_l = bool(np.allclose(losses, g['losses'], rtol=2e-6, atol=0))
        ok_e = bool(np.allclose(users.numpy(), g['final_user_embeds'], rtol=0, atol=2e-6) and
                    np.allclose(items.numpy(), g['final_item_embeds'], rtol=0, atol=2e-6))
        q.put((rank, ok_l, ok_e, len(losses)))
    finally:
        dist.destroy_process_group()


def test_feature_sliced_training_run_reproduces_the_reference_trajectory():
    """the 24-step LightGCN run of the REAL reference (golden traj_tiny_lightgcn) trained on feature-sliced tables with two
    gloo ranks -- sliced parameters, sliced Adam state, the reference's EdgeDrop draws on every rank: same per-step losses,
    same final embeddings (Adam and weight decay are element-wise, so slicing the optimizer state is exact)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_feature_traj_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_l, ok_e, n in res:
        assert n == 24
        assert ok_l, 'rank %d: per-step losses differ from the reference run' % rank
        assert ok_e, 'rank %d: final embeddings differ from the reference run' % rank


class _CpuBipartite:
    """test-side stand-in for a PropGraph of the U x I adjacency: `.mat` = torch sparse A, `.transposed()` = A^T"""

    def __init__(self, mat, other=None):
        self.mat = mat.coalesce()
        self._t = other

    def transposed(self):
        if self._t is None:
            self._t = _CpuBipartite(self.mat.transpose(0, 1), self)
        return self._t


def _feature_lightgcl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        from oracle import ref_expr as R
        from sslrec_amd.data_utils.synth import powerlaw_bipartite
        from sslrec_amd.feature_shard import FeatureSlicedLightGCL, slice_bounds
        U, I, E, d, L, q_rank, temp = 603, 771, 9000, 32, 2, 5, 0.5
        trn = R.binarize_coo(powerlaw_bipartite(U, I, E, seed=13))
        adj = R.lightgcl_adj(trn)
        gen = torch.Generator().manual_seed(2)
        ue, ie = torch.randn(U, d, generator=gen) * 0.1, torch.randn(I, d, generator=gen) * 0.1
        ws = [torch.randn(d, d, generator=gen) * 0.1 for _ in range(L)]
        ut, vt = torch.randn(q_rank, U, generator=gen) * 0.05, torch.randn(q_rank, I, generator=gen) * 0.05
        u_mul_s, v_mul_s = torch.randn(U, q_rank, generator=gen) * 0.05, torch.randn(I, q_rank, generator=gen) * 0.05
        model = FeatureSlicedLightGCL(_CpuBipartite(adj), ue, ie, (ut, vt, u_mul_s, v_mul_s), L, temp, world, rank, device='cpu',
                                      spmm_fn=lambda g, x: torch.sparse.mm(g.mat, x), lowrank_fn=lambda l, r, x: l @ (r @ x))
        B = 53
        batch = [torch.randint(0, U, (B,), generator=gen), torch.randint(0, I, (B,), generator=gen),
                 torch.randint(0, I, (B,), generator=gen)]
        batch[0][:3] = batch[0][3]
        w_params = [w.clone().requires_grad_(True) for w in ws]
        sq = lambda w: w.square().sum()
        loss = model.lightgcl_loss(batch, 0.2, 1e-3, extra_params=w_params, bpr_fn=lambda a, p, n: R.lightgcl_bpr(a, p, n) * B,
                                   reg_fn=sq, infonce_fn=_CpuShardedInfoNceV1.apply)
        loss.backward()
        reg = model.last_parts['reg_local'].clone()
        dist.all_reduce(reg)
        total = model.last_parts['bpr_loss'] + model.last_parts['cl_loss'] + 1e-3 * reg
        rue, rie = ue.clone().requires_grad_(True), ie.clone().requires_grad_(True)
        rws = [w.clone().requires_grad_(True) for w in ws]
        ref_loss, ref_parts = R.lightgcl_cal_loss(adj, rue, rie, rws, (ut, vt, u_mul_s, v_mul_s), batch, L, 1e-3, 0.2, temp)
        ref_loss.backward()
        lo, hi = slice_bounds(d, world, rank)
        ok_f = abs(total.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item())
        ok_f = ok_f and abs(model.last_parts['cl_loss'].item() - ref_parts['cl_loss'].item()) <= 2e-5 * abs(ref_parts['cl_loss'].item())
        ok_b = torch.allclose(model.local_user_embeds.grad, rue.grad[:, lo:hi], rtol=1e-4, atol=1e-7)
        ok_b = ok_b and torch.allclose(model.local_item_embeds.grad, rie.grad[:, lo:hi], rtol=1e-4, atol=1e-7)
        if rank == 0:          # the replicated Ws receive the regularizer's gradient where they are counted
            ok_b = ok_b and all(torch.allclose(w.grad, r.grad, rtol=1e-5, atol=1e-9) for w, r in zip(w_params, rws))
        q.put((rank, bool(ok_f), bool(ok_b), float(total.item()), float(ref_loss.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_feature_sliced_lightgcl_matches_the_oracle_step(world):
    """LightGCL with the tables sliced by embedding column (no collective in the propagation; batch rows of four tables in one
    all-gather; InfoNCE through the slices -> row-blocks transposition): loss parts and gradient columns vs the oracle's step"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_feature_lightgcl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_f, ok_b, total, ref in res:
        assert ok_f, 'rank %d: loss differs (%r vs oracle %r)' % (rank, total, ref)
        assert ok_b, 'rank %d: gradient columns differ from the oracle' % rank
