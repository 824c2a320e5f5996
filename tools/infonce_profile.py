#!/usr/bin/env python
"""The fused InfoNCE call of BASELINE cfg 3's item term (B = 4096 anchors against M = 91,599 rows, d = 64, temp 0.2),
forward + backward, repeated -- the command profiled for profiles/<round>/infonce_*.  usage: python tools/infonce_profile.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = 'cuda:0'
n_item, d, B = 91599, 64, 4096
gen = torch.Generator().manual_seed(0)
t1 = (torch.randn(n_item, d, generator=gen) * 0.1).to(dev).requires_grad_(True)
t2 = (torch.randn(n_item, d, generator=gen) * 0.1).to(dev).requires_grad_(True)
idx = torch.randint(0, n_item, (B,), generator=gen).to(dev)
for _ in range(reps):
    t1.grad = t2.grad = None
    ops.infonce_loss_gathered(t1, t2, idx, 0.2).backward()
torch.cuda.synchronize()
