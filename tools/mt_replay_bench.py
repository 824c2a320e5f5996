"""Time of the device replay of `t.rand(n)` (sslrec_amd.rng.HostGeneratorReplay) for the draw sizes of the BASELINE configs:
EdgeDrop's mask over the amazon-book-shaped graph's entries, one EmbedPerturb table, a whole SimGCL step's 6 tables."""
import json
import sys

import torch

sys.path.insert(0, '.')
from sslrec_amd import rng      # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(1)
import os
if os.environ.get('MT_CFG'):
    a, b, c = (int(v) for v in os.environ['MT_CFG'].split(','))
    rng.HostGeneratorReplay.STRETCH_BLOCKS, rng.HostGeneratorReplay.FAN1, rng.HostGeneratorReplay.FAN2 = a, b, c
rep = rng.enable_host_replay(dev)
rep.draw_ahead = False
out = {}
for name, shape, reps in (('mask_4761460', (4761460,), 20), ('table_144242x64', (144242, 64), 20), ('small_100000', (100000,), 20)):
    rep.rand(shape)             # first call: polynomials, workspace
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x = rep.rand(shape) if not name.startswith('mask') else rep.keep_mask(shape[0], 0.5)
    e1.record()
    torch.cuda.synchronize()
    n = 1
    for s in shape:
        n *= s
    ms = e0.elapsed_time(e1) / reps
    out[name] = {'n': n, 'ms': round(ms, 4), 'numbers_per_s': n / ms * 1e3}
rep.ahead = False
rng.disable_host_replay()
print(json.dumps(out))
