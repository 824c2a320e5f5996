import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/bf16_probe.so'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', os.path.join(here, 'bf16_probe.hip'), '-o', so], check=True)
lib = C.CDLL(so); lib.run_probe.argtypes = [C.c_void_p] * 4
g = torch.Generator().manual_seed(0)
A = torch.randint(-8, 9, (32, 16), generator=g).float().cuda()
B = torch.randint(-8, 9, (16, 32), generator=g).float().cuda()
Cm = torch.zeros(32, 32, device='cuda')
assert lib.run_probe(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
ref = A @ B
print('max abs diff vs A@B:', (Cm - ref).abs().max().item(), ' (0 => assumed operand layout is right)')
