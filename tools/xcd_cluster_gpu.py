#!/usr/bin/env python
"""Locality-aware plan on the GPU (csrc/plan.cpp: cocluster_rows, option / SSLREC_XCD_CLUSTER): the plain column-swept SpMM of one
graph with the rows dealt to the XCDs by load only (cluster 0) or co-clustered (cluster P): time per launch (HIP events), the layout's
distinct (XCD, column) pairs, and -- the two layouts in one process -- bit-identity of the product.  Run it under
`rocprofv3 --pmc FETCH_SIZE` / `TCC_HIT_sum TCC_MISS_sum` with --only P for the fabric traffic of one variant.
usage: python tools/xcd_cluster_gpu.py --graph headline|item_exp1|planted|yelp [--only P] [--reps 50] [--out file.json]"""
import argparse, json, os, sys
import numpy as np, scipy.sparse as sp, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_expr as R
from sslrec_amd import ops
from sslrec_amd.data_utils import synth
from sslrec_amd.graph import PropGraph

ap = argparse.ArgumentParser()
ap.add_argument('--graph', default='headline')
ap.add_argument('--only', type=int, default=None)
ap.add_argument('--passes', type=int, default=4)
ap.add_argument('--reps', type=int, default=50)
ap.add_argument('--out', default=None)
args = ap.parse_args()
u, i, e = synth.SHAPES['amazon-book']
if args.graph == 'headline':
    trn = synth.make_dataset('amazon-book')
elif args.graph == 'item_exp1':
    trn = synth.powerlaw_bipartite(u, i, e, item_exp=1.0)
elif args.graph == 'planted':
    trn = synth.community_bipartite(u, i, e, 64, 0.95)
else:
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'yelp_lightgcn_d64_L2.npz'))
    U, I = (int(x) for x in z['shape'])
    trn = sp.coo_matrix((np.ones(z['trn_row'].size), (z['trn_row'], z['trn_col'])), shape=(U, I))
idx, vals, n = R.normalized_bipartite_coo(R.binarize_coo(trn))
dev, d = 'cuda:0', 64
x = torch.randn(n, d, generator=torch.Generator().manual_seed(0)).to(dev)
res, outs = {'graph': args.graph, 'n_rows': int(n), 'nnz': int(vals.size)}, {}
for P in ([0, args.passes] if args.only is None else [args.only]):
    os.environ['SSLREC_XCD_CLUSTER'] = str(P)
    g = PropGraph(idx[0], idx[1], vals, (n, n), dev)
    lay = g.fwd.swept(d)
    y = ops.spmm_raw(g, x, 'fwd')
    for _ in range(5):
        ops.spmm_raw(g, x, 'fwd', y=y)
    evs = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.spmm_raw(g, x, 'fwd', y=y); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    outs[P] = y.clone()
    res['cluster_%d' % P] = {'launch_us_median': us[len(us) // 2], 'launch_us_min': us[0], 'xcd_col_pairs': lay.xcd_col_pairs,
                             'fabric_floor_MB': lay.xcd_col_pairs * d * 4 / 1e6, 'n_blocks': lay.n_blocks, 'n_slots': lay.n_slots,
                             'hbm_frac_plain': lay.algorithmic_bytes(d) / (us[len(us) // 2] * 1e-6) / 8e12}
    del g
if len(outs) == 2:
    a, b = outs.values()
    res['bit_identical'] = bool(torch.equal(a, b))
print(json.dumps(res), flush=True)
if args.out:
    with open(args.out, 'a') as f:
        f.write(json.dumps(res) + '\n')
