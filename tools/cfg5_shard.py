#!/usr/bin/env python
"""One GPU's share of BASELINE config 5 (LightGCL, 10 M x 10 M interactions graph, d = 128, tables row-sharded over 8 GPUs)
measured on ONE MI355X: rank 0 of 8 generates its cells of the graph (data_utils.synth.sharded_cells: 2/8 of the
interactions), builds its two shard matrices with the native builder (A[my users, :] over the gathered item table and
A^T[my items, :] over the gathered user table, 1.25 M x 10 M each) and times what the rank does locally per step:
the shard products (forward and the mirror-image backward are the same two shapes), the rank-q view and the
all-gather-sized copies that stand in for the exchange's local memory traffic.  Collectives themselves are NOT measured
(one GPU per box); the numbers bound the compute side of sslrec_amd.shard.ShardedLightGCL at full size.
usage: python tools/cfg5_shard.py [--scale 1.0] [--world 8] [--d 128] [--reps 10]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd.data_utils.synth import sharded_cells
from sslrec_amd.graph import PropGraph
from sslrec_amd.shard import gathered_position, rows_per_rank

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--rank', type=int, default=0)
ap.add_argument('--d', type=int, default=128)
ap.add_argument('--q', type=int, default=5)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--degree', type=int, default=32, help='interactions per user of the full graph')
args = ap.parse_args()
P, r, d = args.world, args.rank, args.d
U = I = int(10_000_000 * args.scale)
E = U * args.degree
out = {'workload': 'cfg5 shard', 'n_user': U, 'n_item': I, 'interactions': E, 'world': P, 'rank': r, 'd': d}
t0 = time.time()
(fu, fi), (bu, bi) = sharded_cells(U, I, E, P, r)
out['generate_s'] = round(time.time() - t0, 2)
out['entries_a'], out['entries_at'] = int(fu.size), int(bu.size)
u_per, i_per = rows_per_rank(U, P), rows_per_rank(I, P)
# LightGCL's values need the degrees of BOTH endpoints; the far side's are exchanged at build time in the product
# (ShardedBipartite.from_local_entries).  Timing does not depend on them: the far side gets the mean degree here.
deg_u = np.maximum(np.bincount(fu // P, minlength=u_per), 1).astype(np.float32)
deg_i = np.maximum(np.bincount(bi // P, minlength=i_per), 1).astype(np.float32)
vf = (1.0 / np.sqrt(deg_u[fu // P] * (E / I))).astype(np.float32)
vb = (1.0 / np.sqrt(deg_i[bi // P] * (E / U))).astype(np.float32)
have_gpu = torch.cuda.is_available()
dev = 'cuda:0' if have_gpu else 'cpu'
t0 = time.time()
A = PropGraph._single(fu // P, fi, vf, (u_per, i_per * P), dev, col_relabel=lambda c: gathered_position(c, I, P))
lay_a = A.fwd.swept(d) or A.fwd.packed(d)
out['build_a_s'] = round(time.time() - t0, 2)
t0 = time.time()
AT = PropGraph._single(bi // P, bu, vb, (i_per, u_per * P), dev, col_relabel=lambda c: gathered_position(c, U, P))
lay_at = AT.fwd.swept(d) or AT.fwd.packed(d)
out['build_at_s'] = round(time.time() - t0, 2)
out['kernel'] = type(lay_a).__name__
del fu, fi, bu, bi, vf, vb
if have_gpu:
    from sslrec_amd import ops
    from bench import time_events, HBM_PEAK_GBS
    items_g = torch.randn(i_per * P, d, device=dev)          # the all-gathered item table (5.12 GB at full size)
    users_g = torch.randn(u_per * P, d, device=dev)
    for name, g, x, lay in (('a', A, items_g, lay_a), ('at', AT, users_g, lay_at)):
        y = ops.spmm_raw(g, x, 'fwd')
        out['checksum_' + name] = float(y.double().abs().sum().item())
        ms = time_events(lambda: ops.spmm_raw(g, x, 'fwd'), args.reps, warmup=2)
        nnz = g.fwd.nnz if hasattr(g.fwd, 'nnz') else g.nnz
        alg = lay.algorithmic_bytes(d)
        gathered = nnz * (4 * d + 8) + y.numel() * 4           # every entry gathers a row nobody else holds in cache
        out['spmm_%s_ms' % name] = round(ms, 3)
        out['spmm_%s_algorithmic_GB' % name] = round(alg / 1e9, 3)
        out['spmm_%s_frac_hbm_algorithmic' % name] = round(alg / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
        out['spmm_%s_gathered_GB' % name] = round(gathered / 1e9, 3)
        out['spmm_%s_gather_TBps' % name] = round(gathered / (ms * 1e-3) / 1e12, 3)
        del y
    # the exchange's local side: own shard copied into the gathered buffer + reading it back (RCCL does the rest)
    loc = torch.randn(u_per, d, device=dev)
    ms = time_events(lambda: users_g[:u_per].copy_(loc), args.reps)
    out['exchange_local_copy_ms'] = round(ms, 3)
    out['exchange_bytes_per_rank'] = int(u_per * (P - 1) * d * 4)
    # rank-q SVD view: (u_mul_s_local @ (vt_local @ E_local)) with a q x d all-reduce between the two
    q = args.q
    vt = torch.randn(q, i_per, device=dev); us = torch.randn(u_per, q, device=dev); e_loc = torch.randn(i_per, d, device=dev)
    ms = time_events(lambda: ops.rankq_expand(us, True, ops.rankq_reduce(vt, False, e_loc)), args.reps)
    out['rankq_view_ms'] = round(ms, 3)
    L = 2
    out['products_per_step'] = 4 * L                           # 2 per layer forward + their mirror images backward
    out['local_spmm_ms_per_step'] = round(2 * L * (out['spmm_a_ms'] + out['spmm_at_ms']), 2)
    out['hbm_GB_allocated'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
print(json.dumps(out))
