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


def _cpu_plan_spmm(plan_graph, xg, acc_in, acc_out, want_y):
    """oracle-side stand-in for ops.spmm_raw: walks the SAME packed layout on the host (fp32,
    per-lane-group partial sums added in a fixed order)"""
    sys.path.insert(0, ROOT)
    from tests.helpers import walk_packed
    y = torch.from_numpy(walk_packed(plan_graph.fwd.packed(32), xg.numpy(), dtype=np.float32))
    if acc_out is not None:
        acc_out.copy_(acc_in + y)
    return y if want_y else None


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
        # reduce-scatter dual (column-sharded A): same result up to the order of the P-way fp32 sum
        e0_rs = sg.to_local(e0).requires_grad_(True)
        tot_rs = sharded_propagate_sum(sg, e0_rs, L, spmm_fn=_cpu_plan_spmm, mode='reduce_scatter')
        (tot_rs * sg.to_local(w)).sum().backward()
        ok_f = ok_f and torch.allclose(tot_rs.detach()[:ids.size], tot[ids], rtol=0, atol=1e-5)
        ok_b = ok_b and torch.allclose(e0_rs.grad[:ids.size], g[ids], rtol=0, atol=1e-5)
        ok_f = ok_f and bool((tot_rs.detach()[ids.size:] == 0).all())       # padding rows stay zero
        # full sharded LightGCN step (batch-parallel BPR over all-gathered tables) vs the oracle step
        from sslrec_amd.shard import ShardedGraphCF
        n_user = trn.shape[0]
        sym = PropGraph._single(idx[0], idx[1], vals, (n, n), 'cpu', seg_max=8)     # symmetric graph for the model test
        sgs = ShardedGraph(idx[0], idx[1], vals, n, world, rank, 'cpu', seg_max=8)
        model = ShardedGraphCF(sgs, n_user, n - n_user, e0, 2, spmm_fn=_cpu_plan_spmm)
        B = 37
        batch = [torch.randint(0, n_user, (B,), generator=gen), torch.randint(0, n - n_user, (B,), generator=gen),
                 torch.randint(0, n - n_user, (B,), generator=gen)]
        share = model.lightgcn_loss(batch, 1e-3, bpr_fn=lambda u, i, a, p, q: R.cal_bpr_loss(u[a], i[p], i[q]),
                                    reg_fn=lambda w: w.square().sum())
        share.backward()
        total = share.detach().clone()
        dist.all_reduce(total)
        ue = e0[:n_user].clone().requires_grad_(True); ie = e0[n_user:].clone().requires_grad_(True)
        ref_loss, _ = R.lightgcn_cal_loss(R.torch_adj_from(idx, vals, n), ue, ie, batch, 2, 1.0, 1e-3)
        ref_loss.backward()
        ref_grad = torch.cat([ue.grad, ie.grad])
        ok_f = ok_f and abs(total.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
        ok_b = ok_b and torch.allclose(model.local_embeds.grad[:ids.size], ref_grad[ids], rtol=1e-4, atol=1e-6)
        counts = torch.tensor([sg.nnz_local], dtype=torch.int64)
        dist.all_reduce(counts)
        q.put((rank, bool(ok_f), bool(ok_b), int(counts.item()), int(rows.size)))
    finally:
        dist.destroy_process_group()


def test_sharded_propagation_two_ranks_bitwise_equal_to_one():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
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
