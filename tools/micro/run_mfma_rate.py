import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/mfma_rate.so'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', os.path.join(here, 'mfma_rate.hip'), '-o', so], check=True)
lib = C.CDLL(so); lib.run_rate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
x = torch.randn(8192, device='cuda'); out = torch.zeros(256 * 256, device='cuda')
iters = 20000
for nacc in (1, 2, 4):
    st = torch.cuda.current_stream().cuda_stream
    lib.run_rate(nacc, x.data_ptr(), out.data_ptr(), 100, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.run_rate(nacc, x.data_ptr(), out.data_ptr(), iters, st); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    n = iters * 16
    print('%d accumulator(s): %.1f ns per MFMA per wave  (= %.0f clk at 2.4 GHz); chip %.0f TFLOP/s' % (
        nacc, ms * 1e6 / n, ms * 1e6 / n * 2.4, 256 * 4 * n * 32768 / (ms * 1e-3) / 1e12))
