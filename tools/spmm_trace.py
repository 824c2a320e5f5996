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
# per position of a wave inside its workgroup (waves w, w+4, w+8, w+12 share a SIMD; w is the oldest): when does its sweep end?
wv = np.arange(nw) % 16
t = buf[0].astype(np.float64); t[t == 0] = np.nan; t = (t - np.nanmin(t)) / 100.0
out['sweep_end_by_wave_in_block'] = [round(float(np.nanmedian(t[wv == k, 29])), 1) for k in range(16)]
out['sweep_end_by_simd_age'] = [round(float(np.nanmedian(t[(wv // 4) == q, 29])), 1) for q in range(4)]
# ... and against what the wave's stream contains
steps = lay.w_steps.cpu().numpy()[:nw]
pack = lay.pack.cpu().numpy(); ws = lay.w_start.cpu().numpy()
real = np.array([int((pack[ws[w]:ws[w] + steps[w] // S * 64] != -1).sum()) for w in range(nw)])
se = t[:, 29]
ok = ~np.isnan(se)
out['corr_sweep_end_vs_real_entries'] = round(float(np.corrcoef(se[ok], real[ok])[0, 1]), 3)
out['corr_sweep_end_vs_steps'] = round(float(np.corrcoef(se[ok], steps[ok])[0, 1]), 3)
blk_end = np.array([np.nanmax(se[b * 16:(b + 1) * 16]) for b in range(lay.n_blocks)])
blk_med = np.array([np.nanmedian(se[b * 16:(b + 1) * 16]) for b in range(lay.n_blocks)])
out['block_slowest_minus_median_wave_us'] = {'median': round(float(np.median(blk_end - blk_med)), 1), 'p95': round(float(np.percentile(blk_end - blk_med, 95)), 1)}
out['slowest_wave_position_histogram'] = np.bincount([int(np.nanargmax(se[b * 16:(b + 1) * 16])) for b in range(lay.n_blocks)], minlength=16).tolist()
print(json.dumps(out))
