"""Random-row gather ceiling of the MI355X vector-memory path (tools/micro/gather_ceiling.hip).
usage: python tools/micro/run_gather_ceiling.py [out.json] [--sliced]
--sliced: only the feature-sliced question: 8 per-XCD tables of 144242 rows of d/8 floats, 32-byte
rows (and 64-byte rows = d/4 per XCD pair for comparison); useful TB/s must reach 1.22 GB / 75 us = 16 TB/s to be worth it."""
import ctypes as C, json, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/gather_ceiling.so'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', os.path.join(here, 'gather_ceiling.hip'), '-o', so], check=True)
lib = C.CDLL(so)
lib.launch_gather.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
out = torch.zeros(4096, device='cuda')
st = torch.cuda.current_stream().cuda_stream
res = []
sliced = '--sliced' in sys.argv
if sliced:
    sys.argv.remove('--sliced')
big = '--big' in sys.argv          # tables beyond the L2s: does the 256 MiB Infinity Cache carry narrow-row gathers? (config 5: 10 M rows x 64 B)
if big:
    sys.argv.remove('--big')
for row_bytes in ((64, 256) if big else (32, 64, 256) if sliced else (256, 128, 512, 1024)):
    for T in ([mb * 1000000 // row_bytes for mb in (16, 32, 64, 128, 192, 256, 384, 640, 1280)] if big else (144242, 16384) if sliced else (64, 2048, 8192, 16384, 65536, 144242, 1048576)):
        x = torch.randn(T * row_bytes // 4 * (8 if sliced else 1), device='cuda')
        for threads, blocks, k in (((1024, 512, 8),) if big else ((1024, 256, 8), (1024, 512, 8), (1024, 256, 16)) if sliced else ((1024, 256, 8), (1024, 512, 8), (1024, 256, 4))):
            if k == 4 and row_bytes != 256:
                continue
            if k == 16 and row_bytes != 32:
                continue
            iters = 64
            lpg = row_bytes // 16
            n_gather = blocks * threads // lpg * iters * k

            def run():
                rc = lib.launch_gather(x.data_ptr(), T, row_bytes, iters, blocks, threads, k, out.data_ptr(), st, 1 if sliced else 0)
                assert rc == 0, rc
            for _ in range(3):
                run()
            evs = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); evs.append((e0, e1))
            torch.cuda.synchronize()
            us = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
            tbs = n_gather * row_bytes / us / 1e6
            res.append(dict(row_bytes=row_bytes, table_rows=T, table_MB=T * row_bytes / 1e6, waves_per_cu=blocks * threads // 64 // 256,
                            loads_in_flight=k, per_xcd_tables=bool(sliced), us=us, TBps=tbs, B_per_clk_per_cu=tbs * 1e12 / 256 / 2.4e9))
            print(res[-1], flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
