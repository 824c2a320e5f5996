#!/usr/bin/env python
"""Edge-dropped views on the row-bundled layout at config-5 scale: rank 0's shard matrix A[my 1.25 M users, :] over the
gathered 10 M-row item table (40 M entries) at 16 columns -- the plain product, the zero-valued view of rounds 3-4
(sslrec_bundled_drop_values: full stream length) and the compacted view (sslrec_bundled_compact) at keep rates 0.5 and 0.8, with the
one-off cost of making the view.   usage: python tools/bundled_compact_bench.py [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import time_events  # noqa: E402
from sslrec_amd import ops  # noqa: E402
from sslrec_amd.data_utils.synth import sharded_cells  # noqa: E402
from sslrec_amd.graph import BundledLayout, DroppedView, PropGraph  # noqa: E402
from sslrec_amd.rng import PhiloxState  # noqa: E402

dev, d = 'cuda:0', 16
scale = float(os.environ.get('SCALE', '1.0'))
U = I = int(10_000_000 * scale)
t0 = time.time()
fwd, _ = sharded_cells(U, I, 32 * U, 8, 0)
users, items = fwd
out = {'workload': 'A[my %d users, :] x E_i, %d entries, %d columns, d=%d (row-bundled layout)' % ((U + 7) // 8, users.size, I, d), 'generate_s': round(time.time() - t0, 1)}
vals = np.full(users.size, 0.03, dtype=np.float32)
t0 = time.time()
g = PropGraph._single(users // 8, items, vals, ((U + 7) // 8, I), dev)
lay = g.fwd.packed(d)
assert isinstance(lay, BundledLayout)
out['build_s'] = round(time.time() - t0, 1)
x = torch.randn(I, d, device=dev)
out['plain_ms'] = time_events(lambda: ops.spmm_raw(g, x, 'fwd'), 5, 2)
st = PhiloxState(dev, seed=1)
st.advance()
for keep_rate in (0.5, 0.8):
    rec = {}
    for compacted in (False, True):
        os.environ['SSLREC_BUNDLED_COMPACT'] = '1' if compacted else '0'
        rec['make_view_ms_' + ('compacted' if compacted else 'zero_valued')] = time_events(
            lambda: DroppedView(g, None, 1.0, philox=(st, 1, keep_rate)).compact('fwd', d), 3, 1)
        view = DroppedView(g, None, 1.0, philox=(st, 1, keep_rate))
        arrs = view.compact('fwd', d)
        rec['spmm_ms_' + ('compacted' if compacted else 'zero_valued')] = time_events(lambda: ops.spmm_raw(view, x, 'fwd'), 5, 2)
        if compacted:
            rec['stream_blocks_used_frac'] = float(arrs[3].sum().item()) * 64 / lay.n_elem
            y_c = ops.spmm_raw(view, x, 'fwd')
        else:
            y_z = ops.spmm_raw(view, x, 'fwd')
    rec['bit_identical'] = bool(torch.equal(y_c, y_z))
    out['keep_%.1f' % keep_rate] = rec
os.environ.pop('SSLREC_BUNDLED_COMPACT', None)
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
