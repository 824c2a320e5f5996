#!/usr/bin/env python
"""GPU experiment: both SpMM kernels (column-swept with LDS accumulators / row-streamed) across the synthetic graph
shapes and embedding sizes of the BASELINE configs.  usage: python tools/spmm_kernels.py [--reps 20]"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
from sslrec_amd.graph import PropGraph
from sslrec_amd.data_utils import synth
from bench import time_events

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
dev = 'cuda:0'
for name in ('gowalla', 'yelp', 'amazon-book'):
    trn = synth.make_dataset(name)
    U, I = trn.shape
    n = U + I
    keys = np.unique(trn.row.astype(np.int64) * I + trn.col)
    u, i = keys // I, keys % I + U
    rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    vals = (1.0 / np.sqrt(deg[rows] * deg[cols])).astype(np.float32)
    for d in (32, 64, 128):
        x = torch.randn(n, d, device=dev)
        out = {'graph': name, 'N': n, 'nnz': int(rows.size), 'd': d, 'X_MB': round(n * d * 4 / 1e6, 1)}
        for kernel in ('swept', 'streamed'):
            os.environ['SSLREC_SPMM_SWEPT'] = '1' if kernel == 'swept' else '0'
            g = PropGraph(rows, cols, vals, (n, n), dev)
            if kernel == 'swept' and g.fwd.swept(d) is None:
                out['swept_us'] = None          # output table does not fit the LDS
                continue
            ms = time_events(lambda: ops.spmm_raw(g, x, 'fwd'), args.reps, warmup=3)
            out[kernel + '_us'] = round(ms * 1e3, 1)
            del g
        print(json.dumps(out), flush=True)
