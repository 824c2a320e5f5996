#!/usr/bin/env python
"""runs the cfg-3 item-term InfoNCE (B=4096 x 91,599 rows, d=64) forward+backward a few times; for rocprofv3 --pmc passes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
n_item, d, B = 91599, 64, 4096
t1 = (torch.randn(n_item, d, device='cuda') * 0.1).requires_grad_(True)
t2 = (torch.randn(n_item, d, device='cuda') * 0.1).requires_grad_(True)
idx = torch.randint(0, n_item, (B,), device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t1.grad = t2.grad = None
    ops.infonce_loss_gathered(t1, t2, idx, 0.2).backward()
torch.cuda.synchronize()
