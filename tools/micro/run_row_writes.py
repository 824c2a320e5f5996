"""tools/micro/row_writes.hip: 37 MB of 256-byte rows written by 256 workgroups, by write order.  usage: python tools/micro/run_row_writes.py [out.json]"""
import ctypes as C, json, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/row_writes.so'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', os.path.join(here, 'row_writes.hip'), '-o', so], check=True)
lib = C.CDLL(so)
lib.launch_row_writes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
n_wg, rpw = 256, 563
n = n_wg * rpw
out = torch.zeros(n * 64, device='cuda')
perm = torch.from_numpy(np.random.default_rng(0).permutation(n).astype(np.int32)).cuda()
st = torch.cuda.current_stream().cuda_stream
res = []
for wt in (0, 1):
    for mode, name in ((0, 'rows of a workgroup scattered over the table'), (1, 'contiguous range per workgroup, scattered inside'), (2, 'contiguous range, front to back')):
        def run():
            assert lib.launch_row_writes(out.data_ptr(), perm.data_ptr(), n_wg, rpw, mode, wt, st) == 0
        for _ in range(3):
            run()
        evs = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        us = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
        res.append({'order': name, 'write_through_sc1': bool(wt), 'MB': n * 256 / 1e6, 'us': us, 'TBps': n * 256 / us / 1e6})
        print(res[-1], flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
