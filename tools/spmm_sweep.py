#!/usr/bin/env python
"""GPU experiment: SpMM kernel time across graph sizes and work-list orders (not part of the product).
usage: python tools/spmm_sweep.py [--only amazon-book] [--order degree|xcd] [--reps 20] [--d 64]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
from sslrec_amd.graph import PropGraph
from sslrec_amd.data_utils import synth
from bench import time_events

ap = argparse.ArgumentParser()
ap.add_argument('--only', default=None)
ap.add_argument('--order', default=None)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--d', type=int, default=64)
ap.add_argument('--fold', type=int, default=0, help='map column ids modulo this (L2-resident operand, same row lengths)')
args = ap.parse_args()
synth.SHAPES['l2fit'] = (5000, 7000, 400000)
dev = 'cuda:0'
names = [args.only] if args.only else ['l2fit', 'gowalla', 'yelp', 'amazon-book']
orders = [args.order] if args.order else ['degree', 'xcd']
for name in names:
    trn = synth.make_dataset(name)
    U, I = trn.shape
    n = U + I
    keys = np.unique(trn.row.astype(np.int64) * I + trn.col)
    u, i = keys // I, keys % I + U
    rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    vals = (1.0 / np.sqrt(deg[rows] * deg[cols])).astype(np.float32)
    if args.fold:
        cols = cols % args.fold
    x = torch.randn(n, args.d, device=dev)
    for order in orders:
        g = PropGraph(rows, cols, vals, (n, n), dev, bipartite_split=U if order == 'xcd' else None)
        ms = time_events(lambda: ops.spmm_raw(g, x, 'fwd'), args.reps, warmup=3)
        gather = (g.nnz * (8 + 4 * args.d) + n * args.d * 4) / (ms * 1e-3) / 1e9
        print(json.dumps({'graph': name, 'N': n, 'nnz': int(g.nnz), 'X_MB': n * args.d * 4 / 1e6, 'order': order, 'fold': args.fold,
                          'streams': os.environ.get('SSLREC_SPMM_STREAMS', 'dflt'), 'us': ms * 1e3,
                          'edges_per_s': g.nnz / (ms * 1e-3), 'gather_GBs': gather,
                          'hbm_frac': g.fwd.algorithmic_bytes(args.d) / (ms * 1e-3) / 8e12}), flush=True)
