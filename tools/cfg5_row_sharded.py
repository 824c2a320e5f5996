#!/usr/bin/env python
"""BASELINE config 5 (LightGCL, 10 M x 10 M, 320 M interactions, d = 128, 8 GPUs) in the partition BASELINE.json words -- tables
ROW-SHARDED, one all-gather per product -- as rank 0 of 8 on ONE MI355X: the whole ShardedLightGCL step (reference
models/general_cf/lightgcl.py:73-125) with a COMPLETE breakdown, and what of the exchange can hide under what.

  * every launch of a step attributed: HIP-event brackets around the step's stages in program order (forward) and per autograd
    node (backward hooks), and -- when the command runs under `rocprofv3 --kernel-trace --stats` -- the per-kernel table
    (tools/cfg5_kernel_categories.py sums it by category); the parts are checked against the step time;
  * the exchange: an all-gather hands a rank 7 shards of 640 MB, one per xGMI link.  No link exists on this box, so its wire time
    is a parameter (`--link-gbps`, default 50 and 75 usable per link and direction); what IS measured is whether the compute that
    would run beside it minds the traffic: the InfoNCE kernels and a shard product with 4.48 GB of device copies (the bytes one
    all-gather lands in HBM) running on a second stream;
  * the critical path of the step with the exchange on it, for the serial schedule shard.py runs today and for the overlapped
    schedule the dependencies allow (see `critical_path` in the output), from the measured stage times.

usage: python tools/cfg5_row_sharded.py [--scale 1.0] [--out profiles/r05/cfg5_row_sharded_step.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd import ops, shard as SH  # noqa: E402
from sslrec_amd.data_utils.synth import sharded_cells  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scale', type=float, default=1.0)
ap.add_argument('--d', type=int, default=128)
ap.add_argument('--layers', type=int, default=2)
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--link-gbps', default='50,75')
ap.add_argument('--out', default=None)
ap.add_argument('--modes', default='all_gather,pipelined,separate', help='ShardedLightGCL modes to time (round 6): the fused graph-view node with one all-gather per product, the same with the pipelined per-source-rank exchange, and the separate nodes of rounds 4-5')
args = ap.parse_args()
# the library's default arithmetic of the un-normalized variant (csrc/infonce.hip: inf_precision): h3 since round 6, SSLREC_INFONCE_V1_DEFAULT=x6 restores round 5
V1_PREC = os.environ.get('SSLREC_INFONCE_PRECISION') or ('x6' if (os.environ.get('SSLREC_INFONCE_V1_DEFAULT') or 'h3')[0] == 'x' else 'h3')
U = I = int(10_000_000 * args.scale)
E = U * 32
d, L, B, q, P = args.d, args.layers, args.batch, 5, 8
dev = 'cuda:0'
gen = torch.Generator().manual_seed(2)
mk = lambda r, c, s: (torch.randn(r, c, generator=gen) * s)
out = {'workload': 'cfg5 LightGCL row-sharded: %d x %d, %d interactions, d=%d, L=%d, B=%d; rank 0 of %d on one MI355X' % (U, I, E, d, L, B, P),
       'note': 'collectives replaced by local stand-ins (own shard copied into a persistent gathered buffer; the other ranks\' rows are '
               'random numbers of the same size): wire time is NOT in any measured figure, it enters `critical_path` as a parameter'}

t0 = time.time()
fwd, bwd = sharded_cells(U, I, E, P, 0)
out['generate_s'] = round(time.time() - t0, 1)
SH._all_gather_host = lambda x, world, group=None: np.tile(x, world)                 # degrees of the other ranks' rows: same law
t0 = time.time()
sb = SH.ShardedBipartite.from_local_entries(fwd, bwd, U, I, P, 0, dev)
out['build_s'] = round(time.time() - t0, 1)
out['entries_a'], out['entries_at'] = int(fwd[0].size), int(bwd[0].size)
del fwd, bwd

# ---- stand-in collectives -------------------------------------------------------------------------------------------------------
_gathered = {}
STAGES = []            # (name, start event, end event) of the current step, in issue order
_collect = [False]
_turn = [0]


class stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _collect[0]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if _collect[0]:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            STAGES.append((self.name, self.e0, e1))


def fake_all_gather(x_local, world, group=None, async_op=False):
    """own shard into slot 0 of a persistent [world * n_per, d] buffer whose other slots hold random rows (640 MB copied, not 5 GB)"""
    _turn[0] ^= 1
    key = (tuple(x_local.shape), _turn[0])                      # two buffers per shape, used in turn: a gathered table is read by the launch after it
    with stage('exchange stand-in (own shard copied into the gathered buffer)'):
        buf = _gathered.get(key)
        if buf is None:
            buf = _gathered[key] = torch.randn(world * x_local.shape[0], x_local.shape[1], device=x_local.device) * 0.05
        buf[:x_local.shape[0]].copy_(x_local)
    return (buf, lambda: None) if async_op else buf


SH.all_gather_rows = fake_all_gather
SH.all_reduce_sum = lambda t, group=None: t


def timed(name, fn):
    def wrapped(*a, **k):
        with stage(name):
            return fn(*a, **k)
    return wrapped


# forward stages by wrapping the building blocks; backward stages by wrapping the autograd Functions' backward
SH._default_spmm_orig = SH._default_spmm
spmm_timed = timed('shard product (spmm_stream_kernel<128> + long-row reduce)', SH._default_spmm_orig)
for cls, label in ((SH._ShardedLowRankFn, 'rank-q view'), (ops._InfoNceShardedFn, 'InfoNCE variant 1 (%s)' % V1_PREC), (ops._BprFn, 'BPR variant 1'),
                   (SH._ExchangeRowsFn, 'batch rows (gather + B x d exchange stand-in)'), (ops._SumSqFn, 'regularizer')):
    cls.forward = staticmethod(timed(label + ' fwd', cls.forward))
    cls.backward = staticmethod(timed(label + ' bwd', cls.backward))

fs = 0.05 * (2.0e5 / max(U, 1)) ** 0.5
factors = (mk(q, sb.u_per, fs), mk(q, sb.i_per, fs), mk(sb.u_per, q, 0.05), mk(sb.i_per, q, 0.05))
model = SH.ShardedLightGCL.__new__(SH.ShardedLightGCL)
torch.nn.Module.__init__(model)
model.sb, model.layer_num, model.temp = sb, L, 0.5
model.spmm_fn, model.rankq_fn, model.group = spmm_timed, SH._default_rankq, None
MODES = args.modes.split(',')
model.mode = MODES[0]
model.add_fn = timed('table additions of the fused graph view (sslrec_add_tables_f32)', SH._default_add_tables)
model.lowrank_ops = (timed('rank-q view: reduce (one node per table)', ops.rankq_reduce), timed('rank-q view: expand (one node per table)', ops.rankq_expand))

# pipelined exchange, stand-in: the own shard at once; each of the 7 peers' shards "lands" through a 640 MB device copy on a second
# stream (all seven enqueued up front, one event each -- the shape of `shards_pipelined`'s broadcasts), the block product of source
# rank q waits for q's event.  PIPE_COPIES[0] = False: the peers' rows are simply there (no copies): the compute alone.
PIPE_COPIES = [True]
_peer, _land = {}, {}
_side = torch.cuda.Stream()


def fake_shards_pipelined(x_local, world, rank, group=None):
    shape = tuple(x_local.shape)
    cur = torch.cuda.current_stream()
    evs = {}
    for k in range(1, world):
        qq = (rank + k) % world
        if (shape, qq) not in _peer:
            _peer[(shape, qq)] = torch.randn(*shape, device=x_local.device) * 0.05
            _land[(shape, qq)] = torch.empty(*shape, device=x_local.device)
    if PIPE_COPIES[0]:
        _side.wait_stream(cur)
        with torch.cuda.stream(_side):
            for k in range(1, world):
                qq = (rank + k) % world
                _land[(shape, qq)].copy_(_peer[(shape, qq)], non_blocking=True)
                evs[qq] = torch.cuda.Event()
                evs[qq].record(_side)
    yield rank, x_local
    for k in range(1, world):
        qq = (rank + k) % world
        if PIPE_COPIES[0]:
            cur.wait_event(evs[qq])
            yield qq, _land[(shape, qq)]
        else:
            yield qq, _peer[(shape, qq)]


SH.shards_pipelined = fake_shards_pipelined


def rows_table(n_local, n_per):
    t = torch.zeros(n_per, d)
    t[:n_local] = mk(n_local, d, 0.1)
    return t


model.local_user_embeds = torch.nn.Parameter(rows_table(sb.u_local, sb.u_per).to(dev))
model.local_item_embeds = torch.nn.Parameter(rows_table(sb.i_local, sb.i_per).to(dev))
model.ut, model.vt, model.u_mul_s, model.v_mul_s = (f.to(dev).contiguous() for f in factors)
model.last_parts = {}
batch = [torch.randint(0, U, (B,), generator=gen).to(dev), torch.randint(0, I, (B,), generator=gen).to(dev),
         torch.randint(0, I, (B,), generator=gen).to(dev)]


def step():
    model.local_user_embeds.grad = None
    model.local_item_embeds.grad = None
    loss = model.lightgcl_loss(batch, 0.2, 1e-7)
    loss.backward()
    return loss


def ev_ms(fn, reps, warmup=1):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))


loss = step()
torch.cuda.synchronize()
out['loss'] = float(loss.item())
out['step_ms_compute_only'] = round(ev_ms(step, args.reps), 2)
# ---- the breakdown: one more step with the stage brackets on -----------------------------------------------------------------------
_collect[0] = True
a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a_.record(); step(); b_.record()
torch.cuda.synchronize()
_collect[0] = False
whole = a_.elapsed_time(b_)
stages = [(n, e0.elapsed_time(e1)) for n, e0, e1 in STAGES]
by = {}
for n, ms in stages:
    by.setdefault(n, []).append(round(ms, 3))
bracketed = sum(ms for _, ms in stages)
out['breakdown_ms'] = {n: {'calls': len(v), 'ms_each': v, 'ms': round(sum(v), 2)} for n, v in by.items()}
out['breakdown_ms']['everything else (layer sums `sum(e_u)`, slices, autograd glue: stock elementwise launches on 640 MB tables)'] = {
    'ms': round(whole - bracketed, 2)}
out['breakdown_check'] = {'step_ms_with_brackets': round(whole, 2), 'sum_of_bracketed_stages_ms': round(bracketed, 2),
                          'unbracketed_ms': round(whole - bracketed, 2)}
prod = by['shard product (spmm_stream_kernel<128> + long-row reduce)']
info = sum(sum(v) for n, v in by.items() if n.startswith('InfoNCE'))
info_f = sum(sum(v) for n, v in by.items() if n.startswith('InfoNCE') and n.endswith('fwd'))
rankq = sum(sum(v) for n, v in by.items() if n.startswith('rank-q'))
standin = sum(by.get('exchange stand-in (own shard copied into the gathered buffer)', [0.0]))

# ---- round 6: the other forms of the same step ----------------------------------------------------------------------------------------
out['mode'] = MODES[0]
out['modes'] = {MODES[0]: {'step_ms_compute_only': out['step_ms_compute_only'], 'loss': out['loss']}}
for mode_ in MODES[1:]:
    model.mode = mode_
    rec = {}
    if mode_ == 'pipelined':
        t0 = time.time()
        sb.source_blocks()
        rec['build_source_blocks_s'] = round(time.time() - t0, 1)
        PIPE_COPIES[0] = False
        rec['loss'] = float(step().item())
        rec['step_ms_compute_only'] = round(ev_ms(step, args.reps), 2)
        PIPE_COPIES[0] = True
        step()
        rec['step_ms_with_stand_in_copies_on_a_second_stream'] = round(ev_ms(step, args.reps), 2)
        # the copies alone: 4 L products x 7 shards forward and backward
        shp = (sb.u_per, d)
        srcs = [torch.randn(*shp, device=dev) for _ in range(2)]

        def copies_alone():
            for _ in range(4 * L * 7):
                srcs[1].copy_(srcs[0])
        rec['stand_in_copies_alone_ms'] = round(ev_ms(copies_alone, args.reps), 2)
        rec['stand_in_bytes_per_step'] = 4 * L * 7 * sb.u_per * d * 4
        extra = rec['step_ms_with_stand_in_copies_on_a_second_stream'] - rec['step_ms_compute_only']
        rec['overlap_frac_of_the_stand_in_copies'] = round(1.0 - extra / rec['stand_in_copies_alone_ms'], 3)
        rec['note'] = ('device copies are CU kernels, like RCCL\'s xGMI kernels: overlap_frac = 1 - (step with copies - step without) / copies alone; '
                       'block products of 1/8 of the entries each, own block first')
        del srcs
    else:
        rec['loss'] = float(step().item())
        rec['step_ms_compute_only'] = round(ev_ms(step, args.reps), 2)
    out['modes'][mode_] = rec
model.mode = MODES[0]

# ---- does the compute mind 4.48 GB of copies landing in HBM beside it? --------------------------------------------------------------
side = torch.cuda.Stream()
src = torch.randn(7 * sb.u_per, d, device=dev)
dst = torch.empty_like(src)
e1 = torch.randn(B, d, device=dev) * 0.1
all_local = model.local_user_embeds.detach()[:sb.u_local]


def infonce_fwd():
    with torch.no_grad():
        ops.infonce_loss_sharded(e1, e1, all_local, 0.5, 1, lambda t: t)


x_g = fake_all_gather(model.local_item_embeds.detach(), P)


def product():
    SH._default_spmm_orig(sb.a, x_g, None, None, True)


def beside_copies(fn):
    def run():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dst.copy_(src, non_blocking=True)
        fn()
        torch.cuda.current_stream().wait_stream(side)
    return run


copy_ms = ev_ms(lambda: dst.copy_(src), args.reps)
conc = {'copy_of_4.48GB_alone_ms': round(copy_ms, 3)}
for name, fn in (('infonce_fwd', infonce_fwd), ('shard_product', product)):
    alone = ev_ms(fn, args.reps)
    both = ev_ms(beside_copies(fn), args.reps)
    conc[name] = {'alone_ms': round(alone, 3), 'with_4.48GB_of_copies_on_a_second_stream_ms': round(both, 3),
                  'serial_sum_ms': round(alone + copy_ms, 3)}
out['compute_beside_exchange_traffic'] = conc
del src, dst

# ---- the critical path with the exchange on it -------------------------------------------------------------------------------------
# A list schedule of the step's tasks on two resources -- the GPU (kernels in program order) and the links (all-gathers one after the
# other: each uses all seven links) -- with the dependencies of lightgcl.py:73-125; task durations = the measured stages above, the
# all-gather's = shard bytes / link rate.
def schedule(tasks):
    """tasks: [(name, resource, ms, [deps])] in issue order per resource -> (makespan, {name: (start, end)})"""
    free, done = {}, {}
    for name, res, ms, deps in tasks:
        t0_ = max([free.get(res, 0.0)] + [done[d_][1] for d_ in deps])
        done[name] = (t0_, t0_ + ms)
        free[res] = t0_ + ms
    return max(e for _, e in done.values()), done


def step_tasks(ag, overlapped):
    p_u, p_i = float(np.mean(prod[0::2])), float(np.mean(prod[1::2]))       # A . E_i / A^T . E_u (and their mirror images)
    rq = rankq / (4 * L)                                                     # one rank-q application, forward or backward
    inf_f, inf_b = info_f / 2, (info - info_f) / 2                          # one InfoNCE term
    other = max(compute - (sum(prod) + rankq + info), 0.0)                  # batch rows, BPR, regularizer, layer sums, glue
    link = 'link' if overlapped else 'gpu'                                  # serial: the gather blocks the stream it is issued on
    T = []
    for l in range(1, L + 1):
        dep_i = [] if l == 1 else ['PI%d' % (l - 1)]
        dep_u = [] if l == 1 else ['PU%d' % (l - 1)]
        T += [('AGi%d' % l, link, ag, dep_i), ('AGu%d' % l, link, ag, dep_u)]
        T += [('RQ%d' % l, 'gpu', 2 * rq, dep_i + dep_u), ('PU%d' % l, 'gpu', p_u, ['AGi%d' % l])]
        if l == L and overlapped:
            T += [('INFu_f', 'gpu', inf_f, ['PU%d' % l])]
        T += [('PI%d' % l, 'gpu', p_i, ['AGu%d' % l])]
    if not overlapped:
        T += [('INFu_f', 'gpu', inf_f, ['PU%d' % L])]
    T += [('INFi_f', 'gpu', inf_f, ['PI%d' % L]), ('OTHER', 'gpu', other, ['INFu_f', 'INFi_f'])]
    # backward: user term first, its table gradient is gathered while the item term runs
    T += [('INFu_b', 'gpu', inf_b, ['OTHER']), ('AGdu%d' % L, link, ag, ['INFu_b']), ('INFi_b', 'gpu', inf_b, ['INFu_b']),
          ('AGdi%d' % L, link, ag, ['INFi_b'])]
    for l in range(L, 0, -1):
        T += [('bPU%d' % l, 'gpu', p_i, ['AGdu%d' % l]), ('bPI%d' % l, 'gpu', p_u, ['AGdi%d' % l]), ('bRQ%d' % l, 'gpu', 2 * rq, [])]
        if l > 1:
            T += [('AGdu%d' % (l - 1), link, ag, ['bPI%d' % l]), ('AGdi%d' % (l - 1), link, ag, ['bPU%d' % l])]
    return T


shard_bytes = sb.u_per * d * 4
compute = out['step_ms_compute_only'] - standin           # the stand-in copies are not part of a real step
cp = {'all_gathers_per_step': 4 * L, 'bytes_per_link_per_all_gather': shard_bytes, 'bytes_received_per_rank_per_all_gather': 7 * shard_bytes,
      'compute_ms_without_stand_in_copies': round(compute, 2),
      'dependencies': 'layer l needs AG(E_i^{l-1}) for A.E_i and AG(E_u^{l-1}) for A^T.E_u (two all-gathers share the seven links: one after the '
                      'other); the rank-q views are local; the two InfoNCE terms need the final E_u / E_i (all layers) and are independent of each '
                      'other; backward mirrors it: dE_u is ready after the user term\'s backward, dE_i after the item term\'s',
      'method': 'list schedule of the measured stage times on two resources (GPU in program order, links one all-gather at a time); '
                'serial = the gather blocks the stream (what shard.py issues today), overlapped = gathers on a stream of their own, '
                'issued as soon as their operand exists',
      'schedules': {}}
for gbps in [float(x) for x in args.link_gbps.split(',')]:
    ag = shard_bytes / (gbps * 1e9) * 1e3
    serial, _ = schedule(step_tasks(ag, False))
    overl, when = schedule(step_tasks(ag, True))
    exposed = overl - compute
    cp['schedules']['%g GB/s per link' % gbps] = {
        'ms_per_all_gather': round(ag, 2), 'exchange_ms_if_nothing_overlaps': round(4 * L * ag, 1),
        'step_ms_serial': round(serial, 1), 'step_ms_overlapped': round(overl, 1),
        'exchange_ms_left_on_the_critical_path': round(exposed, 1), 'exchange_hidden_frac': round(1.0 - exposed / (4 * L * ag), 3),
        'timeline_ms_overlapped': {k: [round(v[0], 1), round(v[1], 1)] for k, v in when.items()}}
cp['compare'] = {'feature_sliced_8x16_step_ms': 132.8, 'its_exchange_ms': '~5 (one ~1 MB all-gather + four 80 MB-per-link transpositions)',
                 'source': 'profiles/r03/cfg5_step.json'}
out['critical_path'] = cp
out['hbm_GB_allocated_peak'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
print(json.dumps(out), flush=True)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
