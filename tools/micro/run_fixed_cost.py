"""Fixed cost of the column-swept kernel: the same launch with every stream shortened to 0 steps (LDS zeroing + flush only)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from sslrec_amd import ops, _lib
from sslrec_amd.graph import PropGraph
sys.path.insert(0, os.path.join(ROOT, 'tools'))
for name in ('amazon-book', 'yelp'):
    trn, rows, cols, vals, n = bench.build_graph_host(name)
    for d in (64,):
        g = PropGraph(rows, cols, vals, (n, n), 'cuda')
        lay = g.fwd.swept(d)
        x = torch.randn(n, d, device='cuda')
        y = torch.empty_like(x)
        acc = torch.zeros_like(x)
        zero_steps = torch.zeros_like(lay.w_steps)
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream

        def run(steps, epi=None):
            rc = lib.sslrec_spmm_swept_f32(C.byref(lay.c_struct()), None, None, steps, x.data_ptr(), d, y.data_ptr(),
                                           C.byref(epi) if epi is not None else None, st)
            assert rc == 0
        epi = _lib.EpilogueStruct()
        epi.noise, epi.eps, epi.acc_in, epi.acc_out = None, 0.0, acc.data_ptr(), acc.data_ptr()
        out = {'graph': name, 'd': d, 'rows_per_block': n / 256}
        out['full_us'] = round(bench.time_events(lambda: run(None), 30, 3) * 1e3, 1)
        out['zero_steps_us'] = round(bench.time_events(lambda: run(zero_steps.data_ptr()), 30, 3) * 1e3, 1)
        out['full_acc_us'] = round(bench.time_events(lambda: run(None, epi), 30, 3) * 1e3, 1)
        out['zero_steps_acc_us'] = round(bench.time_events(lambda: run(zero_steps.data_ptr(), epi), 30, 3) * 1e3, 1)
        out['copy_us'] = round(bench.time_events(lambda: y.copy_(x), 30, 3) * 1e3, 1)
        print(out, flush=True)
