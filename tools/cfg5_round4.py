#!/usr/bin/env python
"""BASELINE config 5 (LightGCL, 10 M x 10 M, 320 M interactions, d = 128, 8 GPUs) -- one rank's share of two more decompositions,
measured on ONE MI355X (collectives replaced by local stand-ins of the same size; the bytes a link would carry are printed):

  row_sharded   the partition BASELINE.json words: tables row-sharded (1.25 M users + 1.25 M items per GPU, all 128 columns), one
                all-gather per product; rank 0 of 8 runs a whole ShardedLightGCL step (2 L products + mirror images, rank-q view,
                batch rows, BPR, both staged un-normalized InfoNCE terms) on the row-streamed kernel
  hybrid        2-way rows x 4-way columns: a GPU holds half of the rows and 32 of the 128 columns, so a gather fetches a 128-byte
                row (feature slicing by 8 fetches 64 bytes and is bound by the request rate, profiles/r03/gather_big_tables.json);
                the exchange (an all-gather of the other half of the rows, 640 MB) stays inside pairs.  Products only.

usage: python tools/cfg5_round4.py [--scale 1.0] [--what row_sharded,hybrid] [--out file.json]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd.data_utils.synth import sharded_cells

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--what', default='row_sharded,hybrid')
ap.add_argument('--d', type=int, default=128)
ap.add_argument('--layers', type=int, default=2)
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--out', default=None)
args = ap.parse_args()
U = I = int(10_000_000 * args.scale)
E = U * 32
d, L, B, q = args.d, args.layers, args.batch, 5
dev = 'cuda:0'
from sslrec_amd import ops, shard as SH
from sslrec_amd.graph import PropGraph


def ev_ms(fn, reps, warmup=1):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))


res = {'n_user': U, 'n_item': I, 'interactions': E, 'd': d, 'layers': L, 'batch': B,
       'note': 'collectives replaced by local stand-ins of the same size: wire time NOT included; link figures assume 7 xGMI links of 50-75 GB/s usable each'}
gen = torch.Generator().manual_seed(2)
mk = lambda r, c, s: (torch.randn(r, c, generator=gen) * s)

if 'row_sharded' in args.what:
    P = 8
    out = {'decomposition': 'rows dealt cyclically over 8 GPUs, all %d columns; rank 0' % d}
    t0 = time.time()
    fwd, bwd = sharded_cells(U, I, E, P, 0)
    out['generate_s'] = round(time.time() - t0, 1)
    SH._all_gather_host = lambda x, world, group=None: np.tile(x, world)                 # degrees of the other ranks' rows: same law
    t0 = time.time()
    sb = SH.ShardedBipartite.from_local_entries(fwd, bwd, U, I, P, 0, dev)
    out['build_s'] = round(time.time() - t0, 1)
    out['entries_a'], out['entries_at'] = int(fwd[0].size), int(bwd[0].size)
    del fwd, bwd

    def fake_all_gather(x_local, world, group=None, async_op=False):
        o = x_local.contiguous().repeat(world, 1)
        return (o, lambda: None) if async_op else o
    SH.all_gather_rows = fake_all_gather
    SH.all_reduce_sum = lambda t, group=None: t
    fs = 0.05 * (2.0e5 / max(U, 1)) ** 0.5
    factors = (mk(q, sb.u_per, fs), mk(q, sb.i_per, fs), mk(sb.u_per, q, 0.05), mk(sb.i_per, q, 0.05))

    class _Rows:          # a [U, d] table of which only this rank's rows exist
        def __init__(self, n_local, n_per):
            self.t = torch.zeros(n_per, d)
            self.t[:n_local] = mk(n_local, d, 0.1)
    model = SH.ShardedLightGCL.__new__(SH.ShardedLightGCL)
    torch.nn.Module.__init__(model)
    model.sb, model.layer_num, model.temp = sb, L, 0.5
    model.spmm_fn, model.rankq_fn, model.group = SH._default_spmm, SH._default_rankq, None
    model.local_user_embeds = torch.nn.Parameter(_Rows(sb.u_local, sb.u_per).t.to(dev))
    model.local_item_embeds = torch.nn.Parameter(_Rows(sb.i_local, sb.i_per).t.to(dev))
    model.ut, model.vt, model.u_mul_s, model.v_mul_s = (f.to(dev).contiguous() for f in factors)
    model.last_parts = {}
    batch = [torch.randint(0, U, (B,), generator=gen).to(dev), torch.randint(0, I, (B,), generator=gen).to(dev),
             torch.randint(0, I, (B,), generator=gen).to(dev)]

    def step():
        model.local_user_embeds.grad = None; model.local_item_embeds.grad = None
        loss = model.lightgcl_loss(batch, 0.2, 1e-7)
        loss.backward()
        return loss
    loss = step(); torch.cuda.synchronize()
    out['loss'] = float(loss.item())
    out['step_ms_compute_only'] = round(ev_ms(step, args.reps), 2)
    ops.PROFILE, ops.PROFILE_INFONCE = [], []
    step(); torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    inf, ops.PROFILE_INFONCE = ops.PROFILE_INFONCE, None
    out['spmm_launches_per_step'] = len(prof)
    out['spmm_ms_each'] = [round(a.elapsed_time(b), 3) for a, b, *_ in prof]
    out['spmm_ms_per_step'] = round(sum(out['spmm_ms_each']), 2)
    out['infonce_ms_per_step'] = round(sum(a.elapsed_time(b) for a, b, *_ in inf), 2)
    out['infonce_calls'] = [(r[2], r[3], r[4], r[5]) for r in inf]
    out['kernel'] = type(prof[0][2]).__name__
    n_exch = 4 * L                                          # 2 products per layer forward + their mirror images
    per_rank = (P - 1) / P * (U // P) * d * 4 * P / P       # bytes a rank RECEIVES per exchange from each of 7 peers: one shard each
    shard_bytes = (U // P) * d * 4
    out['exchange'] = {'all_gathers_per_step': n_exch, 'bytes_received_per_rank_per_all_gather': 7 * shard_bytes,
                       'bytes_per_link_per_all_gather': shard_bytes,
                       'ms_per_all_gather_at_50_to_75_GBps_per_link': [round(shard_bytes / 75e9 * 1e3, 1), round(shard_bytes / 50e9 * 1e3, 1)],
                       'ms_per_step_exchange_not_overlapped': [round(n_exch * shard_bytes / 75e9 * 1e3, 1), round(n_exch * shard_bytes / 50e9 * 1e3, 1)],
                       'small_collectives': 'B x d all-reduces (batch rows, InfoNCE row sums / anchor gradients, q x d rank-q sums): < 10 MB per step'}
    out['hbm_GB_allocated_peak'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
    res['row_sharded'] = out
    print(json.dumps({'row_sharded': out}), flush=True)
    del model, sb
    torch.cuda.empty_cache()

if 'hybrid' in args.what:
    out = {'decomposition': '2-way rows x 4-way columns: half of the rows, 32 of %d columns per GPU; rank (0, 0)' % d}
    t0 = time.time()
    fwd, bwd = sharded_cells(U, I, E, 2, 0)                 # interactions of the even users / of the even items
    out['generate_s'] = round(time.time() - t0, 1)
    w = 32
    for name, (rows_g, cols_g), n_r, n_c in (('A[my users, :] @ E_i', fwd, U // 2, I), ('At[my items, :] @ E_u', (bwd[1], bwd[0]), I // 2, U)):
        vals = np.full(rows_g.size, 0.03, dtype=np.float32)
        rec = {'entries': int(rows_g.size)}
        x = torch.randn(n_c, w, device=dev)
        for kern, env in (('row-streamed spmm_stream_kernel<32>', {'SSLREC_SPMM_SWEPT': '0'}), ('row-bundled spmm_bundle_kernel<32>', {'SSLREC_SPMM_SWEPT': '0', 'SSLREC_SPMM_BUNDLED32': '1'})):
            os.environ.update(env)
            t0 = time.time()
            g = PropGraph._single(rows_g // 2, cols_g, vals, (n_r, n_c), dev)
            lay = g.fwd.packed(w)
            b_s = round(time.time() - t0, 1)
            ms = ev_ms(lambda: ops.spmm_raw(g, x, 'fwd'), args.reps)
            gathered = lay.nnz * (4 * w + 8) + lay.n_rows * w * 4
            rec[kern] = {'ms': round(ms, 3), 'build_s': b_s, 'gather_model_TBps': round(gathered / (ms * 1e-3) / 1e12, 2),
                         'hbm_frac_algorithmic': round(lay.algorithmic_bytes(w) / (ms * 1e-3) / 8e12, 4), 'layout': type(lay).__name__}
            for k in env:
                os.environ.pop(k)
            del g, lay
            torch.cuda.empty_cache()
        out[name] = rec
        del x
    half_bytes = (U // 2) * w * 4
    out['exchange'] = {'all_gathers_per_step': 4 * L, 'bytes_per_link_per_all_gather': half_bytes, 'links_used': 1,
                       'ms_per_all_gather_at_50_to_75_GBps': [round(half_bytes / 75e9 * 1e3, 1), round(half_bytes / 50e9 * 1e3, 1)]}
    out['compare'] = {'feature_sliced_8x16_products_ms': [6.2, 5.3], 'source': 'profiles/r03/cfg5_step.json (no exchange in the propagation)'}
    res['hybrid'] = out
    print(json.dumps({'hybrid': out}), flush=True)
if args.out:
    json.dump(res, open(args.out, 'w'), indent=1)
