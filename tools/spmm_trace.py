#!/usr/bin/env python
"""GPU diagnostic: the time line of spmm_swept_kernel inside a LightGCN step, per XCD.  Every wave records the 100 MHz wall
clock at the start of each metadata block, at the end of its sweep, at the start and at the end of its flush
(sslrec_debug_swept_trace keeps the last 4 launches).  Printed per launch and XCD (medians over the XCD's waves, microseconds
after the launch's first time stamp): start of block 0, start of the middle and of the last block, sweep end, flush start,
flush end; plus the p5..p95 spread of the block start times inside an XCD (= how far apart its 32 CUs sweep).
usage: python tools/spmm_trace.py [--graph amazon-book] [--d 64]"""
import argparse, ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import _lib, ops
from sslrec_amd.graph import PropGraph
from bench import build_graph_host, time_events
ap = argparse.ArgumentParser()
ap.add_argument('--graph', default='amazon-book')
ap.add_argument('--d', type=int, default=64)
args = ap.parse_args()
dev = 'cuda:0'
_, rows, cols, vals, n = build_graph_host(args.graph)
g = PropGraph(rows, cols, vals, (n, n), dev)
d, L = args.d, 3
lay = g.fwd.swept(d)
S = lay.steps_per_block(lay.width)
e0 = torch.randn(n, d, device=dev, requires_grad=True)
gt = torch.randn(n, d, device=dev)
lib = _lib.load()
lib.sslrec_debug_swept_trace.argtypes = [C.c_int, C.c_void_p, C.c_int]
nw = lay.n_blocks * 16


def fb():
    e0.grad = None
    ops.propagate_sum(g, e0, L).backward(gt)


ms = time_events(fb, 10, warmup=3)
assert lib.sslrec_debug_swept_trace(1, None, nw) == 0
fb(); fb()
torch.cuda.synchronize()
buf = np.zeros((4, nw, 32), dtype=np.uint64)
launches = lib.sslrec_debug_swept_trace(0, buf.ctypes.data_as(C.c_void_p), nw)
xcd = (np.arange(nw) // 16) % 8
nblk = lay.w_steps.cpu().numpy() // S
out = {'graph': args.graph, 'd': d, 'propagate_fwd_bwd_L3_us': round(ms * 1e3, 1), 'launches_traced': int(launches),
       'blocks_per_wave': [int(nblk.min()), int(nblk.max())], 'launches': []}
for k in range(4):
    t = buf[k].astype(np.float64)
    t[t == 0] = np.nan
    t = (t - np.nanmin(t)) / 100.0
    last = int(nblk.min()) - 1
    rec = {'ring_slot': k, 'xcd': []}
    for x in range(8):
        v = t[xcd == x]
        med = lambda c: round(float(np.nanmedian(v[:, c])), 1)
        spread = lambda c: round(float(np.nanpercentile(v[:, c], 95) - np.nanpercentile(v[:, c], 5)), 1)
        rec['xcd'].append({'x': x, 'block0': med(0), 'block_mid': med(last // 2), 'block_last': med(last), 'sweep_end': med(29),
                           'sweep_end_p95': round(float(np.nanpercentile(v[:, 29], 95)), 1), 'sweep_end_max': round(float(np.nanmax(v[:, 29])), 1),
                           'flush_start': med(30), 'flush_end': med(31), 'flush_end_max': round(float(np.nanmax(v[:, 31])), 1),
                           'spread_mid_p5_p95': spread(last // 2), 'spread_last_p5_p95': spread(last)})
    out['launches'].append(rec)
print(json.dumps(out))
