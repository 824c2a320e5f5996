#!/usr/bin/env python
"""GPU experiment: column-swept SpMM with / without the XCD split of the two row classes of a bipartite adjacency
(SSLREC_SPMM_XCD_SPLIT), on the amazon-book-shaped synthetic graph (BASELINE cfg 2) and on the REAL yelp interactions
(cfg 4; they travel inside tests/golden/yelp_lightgcn_d64_L2.npz), plain / fused-accumulator / edge-dropped (keep 0.5).
usage: python tools/spmm_xcd.py [--reps 30] [--only amazon-book|yelp-real] [--split 0|1] [--d 64]
With --split the script runs ONE configuration (for rocprofv3 --pmc passes)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd import ops
from sslrec_amd.graph import PropGraph, DroppedView
from bench import time_events, build_graph_host


def yelp_real():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'yelp_lightgcn_d64_L2.npz'))
    U, I = (int(v) for v in z['shape'])
    u, i = z['trn_row'].astype(np.int64), z['trn_col'].astype(np.int64) + U
    n = U + I
    rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    vals = (1.0 / np.sqrt(deg[rows] * deg[cols])).astype(np.float32)
    return rows, cols, vals, n


ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=30)
ap.add_argument('--only', default=None)
ap.add_argument('--split', default=None)
ap.add_argument('--d', type=int, default=64)
args = ap.parse_args()
dev = 'cuda:0'
d = args.d
for name in ('amazon-book', 'yelp-real'):
    if args.only and name != args.only:
        continue
    if name == 'yelp-real':
        rows, cols, vals, n = yelp_real()
    else:
        _, rows, cols, vals, n = build_graph_host(name)
    x = torch.randn(n, d, device=dev)
    keep = (torch.rand(rows.size, generator=torch.Generator().manual_seed(1)) + 0.5).floor().bool()
    ref = None
    for split in (('0', '1') if args.split is None else (args.split,)):
        os.environ['SSLREC_SPMM_XCD_SPLIT'] = split
        g = PropGraph(rows, cols, vals, (n, n), dev)
        lay = g.fwd.swept(d)
        out = {'graph': name, 'N': n, 'nnz': int(rows.size), 'd': d, 'xcd_split': bool(lay.xcd_split), 'requested': split,
               'n_slots': lay.n_slots, 'steps_max': int(lay.w_steps.max().item())}
        y = ops.spmm_raw(g, x, 'fwd')
        if ref is None:
            ref = y.clone()
        out['max_abs_diff_vs_first'] = float((y - ref).abs().max().item())
        out['checksum'] = float(y.double().abs().sum().item())
        out['quad_runs'] = int(getattr(lay, 'quad_runs', 0))
        ms = time_events(lambda: ops.spmm_raw(g, x, 'fwd'), args.reps, warmup=3)
        out['plain_us'] = round(ms * 1e3, 2)
        out['plain_frac_hbm'] = round(lay.algorithmic_bytes() / (ms * 1e-3) / 8e12, 4)
        acc = torch.zeros_like(x)
        ms = time_events(lambda: ops.spmm_raw(g, x, 'fwd', acc_in=acc, acc_out=acc, want_y=True), args.reps, warmup=3)
        out['fused_acc_us'] = round(ms * 1e3, 2)
        out['fused_acc_frac_hbm'] = round(lay.algorithmic_bytes(acc=True) / (ms * 1e-3) / 8e12, 4)
        view = DroppedView(g, keep)
        view.masked('fwd', d)
        ms = time_events(lambda: ops.spmm_raw(view, x, 'fwd'), args.reps, warmup=3)
        out['masked_keep0.5_us'] = round(ms * 1e3, 2)
        print(json.dumps(out), flush=True)
        del g, view
