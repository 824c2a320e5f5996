"""Where the eager headline step's time beyond its kernels goes: the same LightGCN step as bench.py's timed region
(a) without any HIP event, (b) with an event pair around every SpMM launch (what bench.py's roofline needs), (c) with a pair around
every n-th launch; and how long the HOST takes to enqueue the K steps (host-bound if that is the whole time)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from sslrec_amd import ops                      # noqa: E402
from sslrec_amd.graph import PropGraph          # noqa: E402


def main():
    dev = 'cuda:0'
    d, L, B, K = 64, 3, 4096, int(os.environ.get('K', '50'))
    trn, rows, cols, vals, n = bench.build_graph_host('amazon-book')
    ue, ie = bench.xavier_tables(trn.shape[0], trn.shape[1], d)
    e0 = torch.cat([ue, ie]).to(dev).requires_grad_(True)
    g = torch.Generator().manual_seed(11)
    batch = [torch.randint(0, trn.shape[0], (B,), generator=g).to(dev), torch.randint(0, trn.shape[1], (B,), generator=g).to(dev),
             torch.randint(0, trn.shape[1], (B,), generator=g).to(dev)]
    graph = PropGraph(rows, cols, vals, (n, n), dev)
    one = torch.ones((), dtype=torch.float32, device=dev)
    ops.SPARSE_GRAD = False

    def step():
        e0.grad = None
        s, reg = ops.propagate_sum(graph, e0, L, reg_weight=1e-8)
        loss, _ = ops.bpr_loss_stacked(s, trn.shape[0], *batch, divisor=B, add=reg)
        loss.backward(one)

    def run(label, every):
        ops.PROFILE = [] if every else None
        if hasattr(ops, 'PROFILE_EVERY'):
            ops.PROFILE_EVERY = every or 1
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        if every:
            ops.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out = {'mode': label, 'ms_per_step': (t2 - t0) / K * 1e3, 'host_enqueue_ms_per_step': (t1 - t0) / K * 1e3}
        if every:
            prof, ops.PROFILE = ops.PROFILE, None
            us = [r[0].elapsed_time(r[1]) * 1e3 for r in prof]
            out.update(event_pairs=len(us), launch_us_mean=sum(us) / len(us))
        return out

    res = []
    for rep in range(2):
        res.append(run('no events', 0))
        res.append(run('event pair around every launch', 1))
        if hasattr(ops, 'PROFILE_EVERY'):
            res.append(run('event pair around every 5th launch', 5))
    for r in res:
        print(json.dumps(r))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'eager_overhead.json'), 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
