"""Prototype: the shipped column-swept layout (lane-group-owned slots, XCD split) with its stream metadata re-packed for
ONE coalesced dword load per 16 steps + DPP row broadcasts (tools/micro/ldsacc2.hip) instead of two 16-byte loads per 4 steps.
usage: python tools/micro/run_dpp.py [workload]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, ROOT)
import bench
from sslrec_amd import ops
from sslrec_amd.graph import PropGraph

workload = sys.argv[1] if len(sys.argv) > 1 else 'amazon-book'
d, G = 64, 4
so2 = '/tmp/ldsacc2.so'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', os.path.join(here, 'ldsacc2.hip'), '-o', so2], check=True)
lib2 = C.CDLL(so2)
P, I = C.c_void_p, C.c_int
lib2.launch_ldsacc2.argtypes = [P] * 10 + [I, I, I, I, P, I, I, P]
trn, rows, cols, vals, n = bench.build_graph_host(workload)
x = torch.randn(n, d, device='cuda')
for split in ('0', '1'):
    os.environ['SSLREC_SPMM_XCD_SPLIT'] = split
    g = PropGraph(rows, cols, vals, (n, n), 'cuda')
    lay = g.fwd.swept(d)
    ref = ops.spmm_raw(g, x, 'fwd')
    ms_ref = bench.time_events(lambda: ops.spmm_raw(g, x, 'fwd'), 30, warmup=3)
    pack, val = lay.pack.cpu().numpy(), lay.val.cpu().numpy()
    w_start, w_steps = lay.w_start.cpu().numpy().astype(np.int64), lay.w_steps.cpu().numpy().astype(np.int64)
    steps16 = -(-w_steps // 16) * 16
    nblk = steps16 // 16
    base = np.cumsum(nblk) - nblk                                  # in 64-dword blocks
    n_elem = int(nblk.sum()) * 64
    p2 = np.full(n_elem, -1, dtype=np.int32)
    v2 = np.zeros(n_elem, dtype=np.float32)
    w_of = np.repeat(np.arange(w_steps.size), w_steps * G)
    k = np.arange(w_of.size) - np.repeat(np.cumsum(w_steps * G) - w_steps * G, w_steps * G)      # position inside the wave's quad stream
    s = (k // (4 * G)) * 4 + k % 4
    gg = (k % (4 * G)) // 4
    src = w_start[w_of] + k
    dst = base[w_of] * 64 + (s // 16) * 64 + gg * 16 + s % 16
    p2[dst] = pack[src]
    v2[dst] = val[src]
    dev = dict(pack=torch.from_numpy(p2).cuda(), val=torch.from_numpy(v2).cuda(), w_start=torch.from_numpy(base.astype(np.int32)).cuda(),
               w_steps=torch.from_numpy(steps16.astype(np.int32)).cuda())
    y = torch.full((n, d), float('nan'), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    dummy = torch.zeros(4, dtype=torch.int32, device='cuda')

    def run(mode=0):
        rc = lib2.launch_ldsacc2(dev['pack'].data_ptr(), dev['val'].data_ptr(), dev['w_start'].data_ptr(), dev['w_steps'].data_ptr(),
                                 x.data_ptr(), y.data_ptr(), lay.f_ptr.data_ptr(), lay.f_row.data_ptr(), lay.f_start.data_ptr(),
                                 lay.f_n.data_ptr(), lay.n_slots, lay.n_blocks, d, mode, dummy.data_ptr(), 0, -1, st)
        assert rc == 0, rc
    run()
    torch.cuda.synchronize()
    err = (y - ref).abs().max().item()
    out = {'split': split, 'shipped_us': round(ms_ref * 1e3, 1), 'dpp_max_abs_diff': err, 'pad_quad': round(1 - lay.nnz / lay.n_elem, 4),
           'pad_dpp': round(1 - lay.nnz / n_elem, 4)}
    for mode, label in ((0, 'dpp_us'), (1, 'dpp_gathers_only_us'), (2, 'dpp_lds_only_us')):
        out[label] = round(bench.time_events(lambda: run(mode), 30, warmup=3) * 1e3, 1)
    print(out, flush=True)
    del g
