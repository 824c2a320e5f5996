#!/usr/bin/env python
"""What a kernel boundary costs between the SpMM launches of a propagation (lever (b) of VERDICT r04 item 2: one persistent launch per
direction would remove L - 1 boundaries and pay L - 1 grid barriers).  The 2 L fused launches of propagate_sum forward + backward
(amazon-book shape, d = 64, L = 3) captured as ONE hipGraph and replayed K times: wall time per replay against the sum of the
launches' own durations by the device clock (first workgroup start -> last workgroup end with its stores drained).  The difference
/ (2 L) is the boundary as the GPU sees it when no host is in the way; the grid barrier that would replace it costs 4.8 - 7.2 us by
MI355X_MICROARCH.md's price list (barrier-xcd), the boundary 1.7 - 1.9 us there."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_graph_host  # noqa: E402
from sslrec_amd import ops  # noqa: E402
from sslrec_amd.graph import PropGraph  # noqa: E402


def main():
    dev, d, L, K = 'cuda:0', 64, 3, 200
    trn, rows, cols, vals, n = build_graph_host('amazon-book')
    graph = PropGraph(rows, cols, vals, (n, n), dev)
    e0 = torch.randn(n, d, device=dev, requires_grad=True)
    gt = torch.randn(n, d, device=dev)

    def fb():
        e0.grad = None
        ops.propagate_sum(graph, e0, L).backward(gt)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fb()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    out = {'workload': 'propagate_sum fwd+bwd, amazon-book shape, d=%d, L=%d: %d fused SpMM launches as one hipGraph' % (d, L, 2 * L),
           'factorized_chain': bool(ops.FACTORIZED)}
    for label, stamped in (('plain', False), ('stamped', True)):
        st = ops.StampLog(dev, 64) if stamped else None
        ops.STAMPS = st
        cg = torch.cuda.CUDAGraph()
        e0.grad = None
        with torch.cuda.graph(cg):
            fb()
        ops.STAMPS = None
        for _ in range(10):
            cg.replay()
        if st is not None:
            st.reset_counts()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            cg.replay()
        torch.cuda.synchronize()
        wall_us = (time.perf_counter() - t0) / K * 1e6
        out['replay_wall_us_' + label] = wall_us
        if st is not None:
            durs = [ms * 1e3 for _, ms, _ in st.read()]
            out['launch_us_by_device_clock'] = durs
            out['sum_of_launches_us'] = float(np.sum(durs))
            out['boundary_us_per_launch'] = (wall_us - float(np.sum(durs))) / len(durs)
        del cg
    print(json.dumps(out))


if __name__ == '__main__':
    main()
