#!/usr/bin/env python
"""One GPU's share of a BASELINE config-5 training STEP with FEATURE-SLICED tables, measured on ONE MI355X:
LightGCL (reference models/general_cf/lightgcl.py:73-125) on the synthetic 10 M x 10 M graph (320 M interactions), d = 128,
8 GPUs -> this rank holds ALL 20 M rows x 16 embedding columns and the WHOLE adjacency (A and A^T in the row-bundled layout,
spmm_bundle_kernel<16>), propagates with no collective, and meets the other ranks only for the batch rows (one all-gather) and
for the un-normalized InfoNCE (one transposition of E_u and E_i to row blocks: this rank then scores the batch against ITS
1.25 M users / items at full width).  The collectives are replaced by local stand-ins that produce tensors of the right size
(own slice tiled P times; sums left as they are): their wire time is NOT measured (one GPU per box) -- everything the rank
COMPUTES per step is: 2 L products forward + their mirror images backward, the rank-q view, batch gathers, BPR, both staged
InfoNCE terms, the scatter of the batch gradients.
usage: python tools/cfg5_step.py [--scale 1.0] [--world 8] [--d 128] [--reps 5]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd.data_utils.synth import cell_bipartite

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--d', type=int, default=128)
ap.add_argument('--q', type=int, default=5)
ap.add_argument('--layers', type=int, default=2)
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--degree', type=int, default=32)
args = ap.parse_args()
P, d, L, B, q = args.world, args.d, args.layers, args.batch, args.q
U = I = int(10_000_000 * args.scale)
E = U * args.degree
w = d // P
out = {'workload': 'cfg5 feature-sliced LightGCL step, one rank of %d' % P, 'n_user': U, 'n_item': I, 'interactions': E, 'd': d,
       'columns_per_gpu': w, 'layers': L, 'batch': B}
t0 = time.time()
cells = [cell_bipartite(U, I, E, P, a, b) for a in range(P) for b in range(P)]
users = np.concatenate([c[0] for c in cells]); items = np.concatenate([c[1] for c in cells])
del cells
out['generate_s'] = round(time.time() - t0, 1)
out['entries'] = int(users.size)
du = np.bincount(users, minlength=U).astype(np.float32); di = np.bincount(items, minlength=I).astype(np.float32)
vals = (1.0 / np.sqrt(du[users] * di[items])).astype(np.float32)          # lightgcl.py:17-20
dev = 'cuda:0'
from sslrec_amd import feature_shard as FS, ops
from sslrec_amd.graph import BundledLayout, PropGraph
t0 = time.time()
graph = PropGraph(users, items, vals, (U, I), dev)
out['csr_build_s'] = round(time.time() - t0, 1)
t0 = time.time()
lay_a = graph.fwd.swept(w) or graph.fwd.packed(w)
lay_at = graph.bwd.swept(w) or graph.bwd.packed(w)
out['layout_build_upload_s'] = round(time.time() - t0, 1)
out['kernel'] = type(lay_a).__name__
out['pads_frac'] = round(1.0 - lay_a.nnz / max(lay_a.n_elem, 1), 4)

# ---- stand-ins for the collectives (sizes as on 8 GPUs, no wire) ---------------------------------------------------------
def fake_all_gather(x_local, world, group=None, async_op=False):
    o = x_local.contiguous().repeat(world, 1)
    return (o, lambda: None) if async_op else o
def fake_all_to_all(send, recv_rows, group):
    return [send[0].new_zeros((int(n), send[0].shape[1])).copy_(send[0][:int(n)]) if send[0].shape[0] >= int(n) else send[0].new_zeros((int(n), send[0].shape[1])) for n in recv_rows]
FS.all_gather_rows = fake_all_gather
FS._all_to_all = fake_all_to_all
FS.all_reduce_sum = lambda t, group=None: t

gen = torch.Generator().manual_seed(2)
mk = lambda r, c, s: (torch.randn(r, c, generator=gen) * s)
ue, ie = mk(U, w, 0.1), mk(I, w, 0.1)                        # this rank's 16 columns (the model slices [lo:hi] of what it is given)
class _Sliced:                                                # [rows, d] tensor of which only the rank's columns exist
    def __init__(self, t, d): self.t, self.shape = t, (t.shape[0], d)
    def __getitem__(self, idx): return self.t
fs = 0.05 * (2.0e5 / max(U, 1)) ** 0.5      # random stand-ins for the SVD factors: scaled so that the rank-q view keeps O(1) rows at any size
factors = (mk(q, U, fs), mk(q, I, fs), mk(U, q, 0.05), mk(I, q, 0.05))
model = FS.FeatureSlicedLightGCL(graph, _Sliced(ue, d), _Sliced(ie, d), factors, L, 0.5, P, 0, device=dev)
batch = [torch.randint(0, U, (B,), generator=gen).to(dev), torch.randint(0, I, (B,), generator=gen).to(dev),
         torch.randint(0, I, (B,), generator=gen).to(dev)]

def step():
    model.local_user_embeds.grad = None; model.local_item_embeds.grad = None
    loss = model.lightgcl_loss(batch, 0.2, 1e-7)
    loss.backward()
    return loss

def ev_ms(fn, reps, warmup=1):
    for _ in range(warmup): fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))

loss = step(); torch.cuda.synchronize()
out['loss'] = float(loss.item())
out['step_ms'] = round(ev_ms(step, args.reps), 2)
ops.PROFILE = []
step(); torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
sp = [(a.elapsed_time(b), plan) for a, b, plan, *_ in prof]
out['spmm_launches_per_step'] = len(sp)
out['spmm_ms_per_step'] = round(sum(t for t, _ in sp), 2)
out['spmm_ms_each'] = [round(t, 3) for t, _ in sp]
alg = lay_a.algorithmic_bytes(w)
out['spmm_algorithmic_GB'] = round(alg / 1e9, 3)
out['spmm_frac_hbm_algorithmic'] = [round(alg / (t * 1e-3) / 8e12, 4) for t, _ in sp]
gathered = lay_a.nnz * (4 * w + 8) + lay_a.n_rows * w * 4
out['spmm_gather_model_GB'] = round(gathered / 1e9, 3)
out['spmm_gather_TBps'] = [round(gathered / (t * 1e-3) / 1e12, 2) for t, _ in sp]
with torch.no_grad():
    e_u, e_i, g_u, g_i = (None,) * 4
    x = model.local_item_embeds.detach()
    out['product_a_ms'] = round(ev_ms(lambda: ops.spmm_raw(graph, x, 'fwd'), args.reps), 3)
    xu = model.local_user_embeds.detach()
    out['product_at_ms'] = round(ev_ms(lambda: ops.spmm_raw(graph, xu, 'bwd'), args.reps), 3)
    out['rankq_view_ms'] = round(ev_ms(lambda: ops.lowrank_apply(model.u_mul_s, model.vt, x), args.reps), 3)
    # parity spot check at full size: 64 sampled rows of A x against an fp64 row product on the host
    y = ops.spmm_raw(graph, x, 'fwd')
    rp, cc, vv = graph.fwd.rowptr_host, graph.fwd.csr_col_host, graph.fwd.csr_val_host
    rows = np.random.default_rng(0).integers(0, U, 64)
    xh = x.cpu().numpy().astype(np.float64)
    err = 0.0
    for r in rows:
        s, e = rp[r], rp[r + 1]
        err = max(err, float(np.abs((vv[s:e, None].astype(np.float64) * xh[cc[s:e]]).sum(0) - y[r].cpu().numpy()).max()))
    out['sampled_rows_max_abs_err_vs_fp64'] = err
out['hbm_GB_allocated_peak'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
out['note'] = 'collectives replaced by local stand-ins of the same size: wire time not included'
print(json.dumps(out))
