import ctypes as C, os, subprocess, sys, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, 'atomic_rows.so')
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-munsafe-fp-atomics', '-fPIC', '-shared', os.path.join(here, 'atomic_rows.hip'), '-o', so], check=True)
lib = C.CDLL(so)
lib.launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
N, n_waves, per_wave = 144242, 5120, 232            # 1.19M row updates = 8 partials x 148k rows
Y = torch.zeros(N, 64, device='cuda')
for name, gen in (('random rows', lambda: torch.randint(0, N, (n_waves * per_wave,), dtype=torch.int32, device='cuda')),):
    rows = gen()
    for mode, label in ((0, 'hip_atomic agent scope'), (1, 'unsafeAtomicAdd'), (2, 'plain store')):
        for _ in range(3):
            lib.launch(Y.data_ptr(), rows.data_ptr(), n_waves, per_wave, mode, torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.launch(Y.data_ptr(), rows.data_ptr(), n_waves, per_wave, mode, torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print('%-14s %-26s %8.1f us per %d row-updates  (%.2f TB/s of 256-B rows)' % (name, label, us, n_waves * per_wave, n_waves * per_wave * 256 / us / 1e6))
