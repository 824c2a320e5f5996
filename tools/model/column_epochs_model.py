#!/usr/bin/env python
"""MODEL, not a measurement: what STATIC COLUMN EPOCHS would cost the column-swept layout in padding.

Idea under test: cut every XCD's column range into K epochs of equal entry mass and pad every lane group's stream to the same
step count per epoch, so that -- without counters or polling -- all lane groups of an XCD are inside the same 1/K of the table at
the same step.  Two numbers decide it before any GPU time is spent:
  * the padding: steps per epoch = the MAX over the XCD's lane groups of their entries in that epoch (lock-step), against the
    mean they run today (streams of equal total length);
  * the prize: tools/model/sweep_l2_model.py replays the sweep through a 4 MiB LRU: today's layout 81.1 % of the gathers hit
    (measured 70-77 %), with NO time drift between waves 83.65 %, with every wave at the same column at the same time 84.5 %
    (profiles/r02/sweep_l2_model.jsonl).  Static epochs remove neither the time drift (memory-latency noise: 2.5 of the 3.4
    points) -- only the column-versus-step misalignment between lane groups: at most 0.85 points = ~10 MB of 278 MB fetched.
usage: python tools/model/column_epochs_model.py [--graph amazon-book] [--d 64]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sslrec_amd.data_utils import synth
from sslrec_amd.graph import PropGraph

ap = argparse.ArgumentParser()
ap.add_argument('--graph', default='amazon-book')
ap.add_argument('--d', type=int, default=64)
args = ap.parse_args()
trn = synth.make_dataset(args.graph)
U, I = trn.shape
n = U + I
keys = np.unique(trn.row.astype(np.int64) * I + trn.col)
u, i = keys // I, keys % I + U
rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
deg = np.bincount(rows, minlength=n).astype(np.float64)
vals = (1.0 / np.sqrt(deg[rows] * deg[cols])).astype(np.float32)
lay = PropGraph(rows, cols, vals, (n, n), 'cpu').fwd.swept(args.d)
G = lay.G
S, LPG = lay.steps_per_block(lay.width), 64 // G
pack, ws, wst = lay.pack.numpy(), lay.w_start.numpy(), lay.w_steps.numpy()
out = {'graph': args.graph, 'd': args.d, 'lane_groups_per_wave': G, 'steps_per_wave_today': int(wst.max()), 'by_K': {}}
for x in (0, 4):                                     # one XCD of each row class
    waves = [w for w in range(lay.n_blocks * 16) if (w // 16) % 8 == x]
    streams = []                                     # columns of every lane group of the XCD, in stream order
    for w in waves:
        steps = int(wst[w])
        s = np.arange(steps)
        for g in range(G):
            pk = pack[ws[w] + (s // S) * 64 + g * LPG + s % S]
            streams.append((pk[pk != -1].view(np.uint32) & 0xFFFFF).astype(np.int64))
    allc = np.sort(np.concatenate(streams))
    today = max(len(c) for c in streams)
    for K in (4, 8, 16, 32):
        edges = allc[(np.arange(1, K) * allc.size) // K]                  # epoch boundaries: equal entry mass
        cnt = np.stack([np.bincount(np.searchsorted(edges, c, side='right'), minlength=K) for c in streams])    # [lane groups, K]
        padded = int(cnt.max(0).sum())                                    # lock-step: every epoch as long as its longest lane group
        rec = out['by_K'].setdefault(str(K), {})
        rec['xcd%d' % x] = {'steps_today': today, 'steps_with_epochs': padded, 'padding_overhead': round(padded / today - 1.0, 3),
                            'epoch_MB_of_the_gathered_table': round((allc.max() - allc.min() + 1) * args.d * 4 / K / 1e6, 2)}
print(json.dumps(out, indent=1))
