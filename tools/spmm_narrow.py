#!/usr/bin/env python
"""GPU experiment: one rank's share of the feature-sliced propagation (sslrec_amd/feature_shard.py) on the amazon-book-
shaped graph -- the column-swept SpMM at d / P = 64, 32, 16, 8 columns: plain product, and the six fused launches of a
LightGCN step (propagate_sum forward + backward).  usage: python tools/spmm_narrow.py [--reps 20] [--graph amazon-book]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
from sslrec_amd.graph import PropGraph
from sslrec_amd.data_utils import synth
from bench import time_events

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--graph', default='amazon-book')
ap.add_argument('--widths', default='64,32,16,8')
args = ap.parse_args()
dev = 'cuda:0'
trn = synth.make_dataset(args.graph)
U, I = trn.shape
n = U + I
keys = np.unique(trn.row.astype(np.int64) * I + trn.col)
u, i = keys // I, keys % I + U
rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
deg = np.bincount(rows, minlength=n).astype(np.float64)
vals = (1.0 / np.sqrt(deg[rows] * deg[cols])).astype(np.float32)
g = PropGraph(rows, cols, vals, (n, n), dev)
L = 3
for w in [int(t) for t in args.widths.split(',')]:
    lay = g.fwd.swept(w)
    x = torch.randn(n, w, device=dev)
    out = {'graph': args.graph, 'nnz': int(rows.size), 'width': w, 'ranks_at_d64': 64 // w, 'n_slots': lay.n_slots,
           'pads_frac': round(1.0 - rows.size / (lay.n_elem / max(1, (64 // lay.G) // 16 if lay.G < 4 else 1)), 4),
           'max_chunks_per_row': int(lay.f_n.max())}
    torch.manual_seed(0)
    x = torch.randn(n, w, device=dev)
    out['checksum'] = float(ops.spmm_raw(g, x, 'fwd').double().abs().sum().item())
    ms = time_events(lambda: ops.spmm_raw(g, x, 'fwd'), args.reps, warmup=3)
    out['plain_us'] = round(ms * 1e3, 1)
    e0 = torch.randn(n, w, device=dev, requires_grad=True)
    gt = torch.randn(n, w, device=dev)

    def fb():
        e0.grad = None
        ops.propagate_sum(g, e0, L).backward(gt)
    ms = time_events(fb, args.reps, warmup=3)
    out['propagate_fwd_bwd_L3_us'] = round(ms * 1e3, 1)
    out['edges_per_s_rank'] = round(2 * L * rows.size / (ms * 1e-3))
    print(json.dumps(out), flush=True)
