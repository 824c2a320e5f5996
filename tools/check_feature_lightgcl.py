#!/usr/bin/env python
"""GPU check still to be run (no GPU time was left in round 2): FeatureSlicedLightGCL with the REAL kernels, `--world` processes
on one GPU (gloo collectives, host-staged), against the oracle's LightGCL step on the whole graph -- the GPU counterpart of
tests/test_shard_gloo.py::test_feature_sliced_lightgcl_matches_the_oracle_step.  Prints one JSON line per rank; once it passes it
belongs in tests/test_gpu_parity.py.   usage: python tools/check_feature_lightgcl.py [--world 2] [--d 64]"""
import argparse, json, os, socket, sys
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, d, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import ref_expr as R
        from sslrec_amd.data_utils.synth import powerlaw_bipartite
        from sslrec_amd.feature_shard import FeatureSlicedLightGCL, slice_bounds
        from sslrec_amd.graph import PropGraph
        dev = 'cuda:0'
        U, I, E, L, q_rank, temp, B = 603, 771, 9000, 2, 5, 0.5, 53
        trn = R.binarize_coo(powerlaw_bipartite(U, I, E, seed=13))
        adj = R.lightgcl_adj(trn).coalesce()
        idx, vals = adj.indices().numpy(), adj.values().numpy()
        gen = torch.Generator().manual_seed(2)
        ue, ie = torch.randn(U, d, generator=gen) * 0.1, torch.randn(I, d, generator=gen) * 0.1
        ut, vt = torch.randn(q_rank, U, generator=gen) * 0.05, torch.randn(q_rank, I, generator=gen) * 0.05
        u_mul_s, v_mul_s = torch.randn(U, q_rank, generator=gen) * 0.05, torch.randn(I, q_rank, generator=gen) * 0.05
        batch = [torch.randint(0, U, (B,), generator=gen), torch.randint(0, I, (B,), generator=gen),
                 torch.randint(0, I, (B,), generator=gen)]
        graph = PropGraph(idx[0], idx[1], vals, (U, I), dev)
        model = FeatureSlicedLightGCL(graph, ue, ie, (ut, vt, u_mul_s, v_mul_s), L, temp, world, rank)
        loss = model.lightgcl_loss([b.to(dev) for b in batch], 0.2, 1e-3)
        loss.backward()
        reg = model.last_parts['reg_local'].clone().cpu()
        dist.all_reduce(reg)
        total = model.last_parts['bpr_loss'].item() + model.last_parts['cl_loss'].item() + 1e-3 * reg.item()
        rue, rie = ue.clone().requires_grad_(True), ie.clone().requires_grad_(True)
        ref_loss, ref_parts = R.lightgcl_cal_loss(adj, rue, rie, [], (ut, vt, u_mul_s, v_mul_s), batch, L, 1e-3, 0.2, temp)
        ref_loss.backward()
        lo, hi = slice_bounds(d, world, rank)
        gu = (model.local_user_embeds.grad.cpu() - rue.grad[:, lo:hi]).abs().max().item() / rue.grad.abs().max().item()
        gi = (model.local_item_embeds.grad.cpu() - rie.grad[:, lo:hi]).abs().max().item() / rie.grad.abs().max().item()
        q.put({'rank': rank, 'width': hi - lo, 'loss': total, 'oracle': ref_loss.item(), 'loss_rel_err': abs(total - ref_loss.item()) / abs(ref_loss.item()),
               'grad_user_rel_err': gu, 'grad_item_rel_err': gi, 'ok': bool(abs(total - ref_loss.item()) <= 2e-5 * abs(ref_loss.item()) and gu < 1e-4 and gi < 1e-4)})
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--d', type=int, default=64)
    args = ap.parse_args()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.world, port, args.d, q)) for r in range(args.world)]
    for p in procs:
        p.start()
    for _ in procs:
        print(json.dumps(q.get(timeout=600)))
    for p in procs:
        p.join(timeout=60)
