#!/usr/bin/env python
"""N eager training steps (cal_loss + backward, perf-mode RNG) of one BASELINE config through the model classes and nothing else -- the
command `rocprofv3 --kernel-trace --stats` is pointed at for the per-kernel tables of profiles/r05/cfg{3,4}_kernel_stats.csv (bench.py
--config also times other arithmetic modes and a captured graph, which would mix into the table).
usage: python tools/step_profile.py cfg3|cfg4|cfg1 [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_configs import _build  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = 'cuda:0'
trn, dh, model, mcfg = _build(tag, dev, device_rng=True)
gen = torch.Generator().manual_seed(1)
B = 4096
batch = [torch.randint(0, trn.shape[0], (B,), generator=gen).to(dev), torch.randint(0, trn.shape[1], (B,), generator=gen).to(dev),
         torch.randint(0, trn.shape[1], (B,), generator=gen).to(dev)]
for _ in range(steps):
    model.zero_grad(set_to_none=True)
    loss, _ = model.cal_loss(batch)
    loss.backward()
torch.cuda.synchronize()
print('%s: %d steps, last loss %.6f' % (tag, steps, loss.item()))
