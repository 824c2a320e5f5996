"""The single-GPU configurations of BASELINE.json besides the headline (cfg 2 is bench.py's own line) as bench lines:

  cfg1  LightGCN step on the gowalla-shaped synthetic graph (the train pickle is missing upstream), d=32, L=2, keep_rate 1.0
  cfg3  SimGCL step on the amazon-book-shaped graph, d=64, L from simgcl.yml (2)      reference: models/general_cf/simgcl.py:39-55
  cfg4  SGL-ED step on the REAL yelp interactions (tests/golden/yelp_lightgcn_d64_L2.npz), d=64, L from sgl.yml (2), keep 0.5
                                                                                        reference: models/general_cf/sgl.py:45-65

A step = the model class's cal_loss + backward at B = 4096.  Every line carries: the step time with the augmentation randomness
computed in the kernels (perf mode, `model.device_rng`), with the reference's own CPU generator continued on the device (parity mode,
sslrec_amd.rng.HostGeneratorReplay), and replayed as ONE captured hipGraph; the roofline of the kernel that DOMINATES the step -- the
column-swept SpMM against HBM for cfg 1, the fused InfoNCE against the bf16 MFMA peak for cfg 3 / cfg 4 -- from HIP events around every
launch / call inside the timed region; the SpMM's HBM figure beside it; and `cpu_baseline`: the oracle's restatement of the same
reference step (oracle/ref_expr.py) on the host cores, same graph, same batch, a bounded number of steps.
`python bench.py` runs all three after its headline (`configs` in its line); `python bench.py --config cfgN` prints one as a line."""
import json
import math
import os
import time

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3


def yelp_real():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'yelp_lightgcn_d64_L2.npz'))
    U, I = (int(v) for v in z['shape'])
    return sp.coo_matrix((np.ones(z['trn_row'].size, dtype=np.float32), (z['trn_row'], z['trn_col'])), shape=(U, I))


# tag -> (model, graph, d, layers (None = the model yml's), extra model config, description)
CONFIGS = {
    'cfg1': ('lightgcn', 'gowalla', 32, 2, {'keep_rate': 1.0}, 'LightGCN on the gowalla-shaped synthetic graph, d=32, L=2'),
    'cfg3': ('simgcl', 'amazon-book', 64, None, {}, 'SimGCL on the amazon-book-shaped synthetic graph, d=64 (uniform-noise views + InfoNCE)'),
    'cfg4': ('sgl', 'yelp-real', 64, None, {'keep_rate': 0.5}, 'SGL-ED on the real yelp interactions, d=64, keep 0.5 (two edge-dropped views + InfoNCE)'),
}


def _build(tag, dev, device_rng):
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
    from sslrec_amd.data_utils.synth import make_dataset
    from sslrec_amd.models.bulid_model import build_model
    model_name, graph_name, d, L, extra, _ = CONFIGS[tag]
    raw = yelp_real() if graph_name == 'yelp-real' else make_dataset(graph_name)
    trn = sp.coo_matrix((raw != 0).astype(np.float32))          # what DataHandlerGeneralCF._load_one_mat does to a pickle
    over = {'data': {'synthetic': 'tiny'}, 'model': {'embedding_size': d, 'device_rng': bool(device_rng)}}
    if L is not None:
        over['model']['layer_num'] = L
    over['model'].update(extra)
    load_config(model_name, device=dev, overrides=over)
    dh = DataHandlerGeneralCF()
    dh.trn_mat = trn
    configs['data']['user_num'], configs['data']['item_num'] = trn.shape
    dh.torch_adj = dh._make_torch_adj(trn).to(dev)
    torch.manual_seed(0)
    model = build_model(dh).to(dev)
    return trn, dh, model, dict(configs['model'])


def _timed(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _cpu_step_baseline(tag, trn, mcfg, batch, budget_s):
    """the oracle's restatement of the reference step (cal_loss + backward) on the host cores: same graph, same batch; returns
    (fastest step in seconds, steps run, thread count of the fastest)"""
    from oracle import ref_expr as R
    model_name = CONFIGS[tag][0]
    idx, vals, n = R.normalized_bipartite_coo(trn)
    adj = R.torch_adj_from(idx, vals, n)
    d, L = mcfg['embedding_size'], mcfg['layer_num']
    gen = torch.Generator().manual_seed(0)
    ue = (torch.rand(trn.shape[0], d, generator=gen) - 0.5).mul_(0.02).requires_grad_(True)
    ie = (torch.rand(trn.shape[1], d, generator=gen) - 0.5).mul_(0.02).requires_grad_(True)
    cb = [b.cpu() for b in batch]

    def step():
        ue.grad = ie.grad = None
        if model_name == 'lightgcn':
            loss, _ = R.lightgcn_cal_loss(adj, ue, ie, cb, L, mcfg['keep_rate'], mcfg['reg_weight'])
        elif model_name == 'simgcl':
            loss, _ = R.simgcl_cal_loss(adj, ue, ie, cb, L, mcfg['reg_weight'], mcfg['cl_weight'], mcfg['temperature'], mcfg['eps'])
        else:
            loss, _ = R.sgl_cal_loss(adj, ue, ie, cb, L, mcfg['keep_rate'], mcfg['reg_weight'], mcfg['cl_weight'], mcfg['temperature'])
        loss.backward()
    # torch's CPU kernels for these shapes get SLOWER with hundreds of threads (the GPU box has 256 cores: 45 s per SimGCL step with
    # all of them): the baseline is the best of a few thread counts, each tried while the budget lasts, and says which it used
    best, best_thr, n_steps = None, None, 0
    t_all = time.perf_counter()
    for thr in sorted({min(16, os.cpu_count()), min(64, os.cpu_count()), os.cpu_count()}):
        if best is not None and time.perf_counter() - t_all + 1.5 * best > budget_s:
            break
        torch.set_num_threads(thr)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        n_steps += 1
        if best is None or dt < best:
            best, best_thr = dt, thr
        if time.perf_counter() - t_all + dt < budget_s:      # a second step at this thread count (the first pays first-touch costs)
            t0 = time.perf_counter()
            step()
            dt = time.perf_counter() - t0
            n_steps += 1
            if dt < best:
                best, best_thr = dt, thr
    torch.set_num_threads(os.cpu_count())
    return float(best), n_steps, best_thr


def epoch_times(dev='cuda:0', model_name='lightgcn', epochs=2):
    """What ONE EPOCH of `Trainer.train_epoch` (reference trainer/trainer.py:51-84: sample_negs, the shuffled loader, per step zero_grad ->
    cal_loss -> .item() -> backward -> Adam) costs at amazon-book size (582 steps of B = 4096, d = 64, L = 3) -- the figure a user of
    `python main.py --model lightgcn` sees, which no kernel line shows (VERDICT r05 missing #5):
      default   every default of the package: the reference's batches and draws bit for bit (native MT19937 negative sampler, array-slicing
                loader, EdgeDrop's CPU generator continued on the device), eager launches, torch's Adam;
      python    the same numbers through the reference's own host statements (train.python_neg_sampling + train.torch_dataloader): what
                rounds 1-5 ran by default;
      default_graphed  the default's batches and draws (still the reference's, bit for bit) with the step replayed as a hipGraph and the fused
                Adam: parameters within 1e-6 of the eager run (tests/test_gpu_parity.py: test_hip_graph_training_in_parity_mode_...);
      fast      every opt-in: device sampler + device loader, Philox augmentation in the kernels, the step replayed as a hipGraph, fused Adam
                (a different random stream: statistical, not bitwise, parity).
    Returns seconds per epoch (the median of `epochs` epochs after one warm-up epoch) and the split of the default epoch."""
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.build_data_handler import build_data_handler
    from sslrec_amd.models.bulid_model import build_model
    from sslrec_amd.trainer.logger import Logger
    from sslrec_amd.trainer.trainer import Trainer, init_seed
    from sslrec_amd import rng as rng_mod
    out = {'model': model_name, 'graph': 'amazon-book-shaped synthetic (52643 x 91599, 2,380,730 interactions)', 'd': 64, 'L': 3, 'B': 4096}
    cwd = os.getcwd()
    os.makedirs('/tmp/sslrec_epoch', exist_ok=True)
    os.chdir('/tmp/sslrec_epoch')
    try:
        for tag, train_over, model_over, opt_over in (
                ('default', {}, {}, {}),
                ('python', {'python_neg_sampling': True, 'torch_dataloader': True}, {}, {}),
                ('default_graphed', {'hip_graph': True}, {}, {'fused': True}),
                ('fast', {'fast_loader': True, 'device_sampler': True, 'hip_graph': True}, {'device_rng': True}, {'fused': True})):
            load_config(model_name, device=dev, overrides={
                'data': {'synthetic': 'amazon-book'},
                'train': dict({'epoch': 1, 'log_loss': False, 'save_model': False, 'reproducible': True, 'seed': 2023}, **train_over),
                'model': dict({'embedding_size': 64, 'layer_num': 3}, **model_over), 'optimizer': opt_over})
            init_seed()
            dh = build_data_handler()
            dh.load_data()
            model = build_model(dh).to(dev)
            trainer = Trainer(dh, Logger(log_configs=False))
            trainer.create_optimizer(model)
            replay = str(dev).startswith('cuda') and not model_over.get('device_rng')
            if replay:
                rng_mod.enable_host_replay(torch.device(dev))
            try:
                n = 1 if tag == 'python' else epochs          # (the Python sampler alone is ~5 s per epoch)
                times = []
                for ep in range(n + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    trainer.train_epoch(model, ep)
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - t0)
                out['epoch_s_' + tag] = float(np.median(times[1:]))
                out['first_epoch_s_' + tag] = times[0]
                if tag == 'default':      # where the default epoch's host time goes
                    ds = dh.train_dataloader.dataset
                    t0 = time.perf_counter(); ds.sample_negs(); out['default_sample_negs_s'] = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    nb = sum(1 for _ in dh.train_dataloader)
                    torch.cuda.synchronize()
                    out['default_loader_s'] = time.perf_counter() - t0
                    out['steps_per_epoch'] = nb
            finally:
                if replay:
                    rng_mod.disable_host_replay()
            del trainer, model, dh
            torch.cuda.empty_cache()
    finally:
        os.chdir(cwd)
    if out.get('epoch_s_fast'):
        out['default_over_fast'] = out['epoch_s_default'] / out['epoch_s_fast']
    return out


HEADLINE_FORM = {'cfg1': 'graph', 'cfg3': 'eager', 'cfg4': 'eager'}


def run_config(tag, steps=30, warmup=5, dev='cuda:0', cpu_budget_s=10.0, with_cpu=True, with_parity_mode=True):
    from sslrec_amd import ops
    from sslrec_amd import rng as rng_mod
    model_name, graph_name, d, _, _, desc = CONFIGS[tag]
    trn, dh, model, mcfg = _build(tag, dev, device_rng=True)
    L = mcfg['layer_num']
    B = 4096
    gen = torch.Generator().manual_seed(1)
    batch = [torch.randint(0, trn.shape[0], (B,), generator=gen).to(dev), torch.randint(0, trn.shape[1], (B,), generator=gen).to(dev),
             torch.randint(0, trn.shape[1], (B,), generator=gen).to(dev)]

    def step():
        model.zero_grad(set_to_none=True)
        loss, _ = model.cal_loss(batch)
        loss.backward()
    for _ in range(warmup):
        step()
    # an event pair around every n-th SpMM launch (n coprime with the launches of a step: the sampled launch rotates through them) --
    # a pair around every launch costs the stream a few microseconds of bubbles each (tools/eager_overhead.py)
    ops.PROFILE, ops.PROFILE_EVERY = [], 1
    step()
    every = 5
    while math.gcd(every, max(1, len(ops.PROFILE))) != 1:
        every += 2
    if steps * len(ops.PROFILE) < 4 * every:      # (a run of a few steps: every launch)
        every = 1
    ops.PROFILE, ops.PROFILE_EVERY, ops.PROFILE_INFONCE = [], every, []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    all_launches, ops.PROFILE, ops.PROFILE_EVERY = ops.PROFILE, None, 1
    prof = [r for r in all_launches if r[0] is not None]
    inf, ops.PROFILE_INFONCE = ops.PROFILE_INFONCE, None
    # the same step as ONE captured hipGraph (what `train.hip_graph` does for the Trainer): launch-bound steps -- cfg 1's 17 small
    # launches take 0.12 ms of GPU time and 0.4 ms of Python -- show what the device is left with
    graph_ms, graph_err = None, None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            loss_g, _ = model.cal_loss(batch)
            loss_g.backward()
        graph_ms = _timed(g.replay, steps, 3)
        del g
    except Exception as exc:      # (a step that cannot be captured keeps its eager line)
        graph_err = repr(exc)[:300]
    # SpMM launches: HBM roofline on the algorithmic bytes of SURVEY.md 8d (a launch told which rows of its operand are zero --
    # LightGCN's first backward product -- reads only those; the last forward launch of a deferred layer sum reads the earlier layers)
    k_ms = [a.elapsed_time(b) for a, b, *_ in prof]
    k_bytes = [r[2].algorithmic_bytes(r[3], acc=r[4], write_y=r[5], **({'x_rows': r[7]} if len(r) > 7 and r[7] is not None else {}),
                                      **({'sum_in': r[8]} if len(r) > 8 and r[8] else {}))
               - (1.0 - r[6]) * r[2].nnz * 8 for r in prof]
    # a launch with K epilogues (SimGCL's three views share their first product) writes K outputs and K running sums where the
    # formula above counts one of each: + (K - 1) (Y + acc_out) -- the shared operand, the entries and acc_in (E0) are read once
    def _meta(r):
        return r[10] if len(r) > 10 and isinstance(r[10], dict) else {'views': 1, 'perturbed': False, 'philox': False, 'which': '?'}
    k_bytes = [b + (_meta(r)['views'] - 1) * r[2].n_rows * r[3] * 4 * ((1 if r[5] else 0) + (1 if r[4] else 0)) for b, r in zip(k_bytes, prof)]
    edges = float(np.sum([r[2].nnz * r[6] for r in all_launches])) / steps
    spmm_ach = float(np.sum(k_bytes)) / (float(np.sum(k_ms)) * 1e-3) / 1e9
    spmm_ms_step = float(np.mean(k_ms)) * len(all_launches) / steps
    spmm = {'bound': 'hbm', 'achieved': spmm_ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': spmm_ach / HBM_PEAK_GBS, 'traffic': None,
            'kernel': 'spmm_swept_kernel<%d> (LDS accumulators, column-swept)' % d if type(prof[0][2]).__name__ == 'SweptLayout' else type(prof[0][2]).__name__,
            'avg_launch_us': float(np.mean(k_ms)) * 1e3, 'launches': len(all_launches), 'launches_timed': len(prof), 'ms_per_step': spmm_ms_step,
            'launch_timing': 'HIP events around every SpMM launch of the timed region' if every == 1 else
            'HIP events around every %dth SpMM launch of the timed region (the sampled launch rotates through the %d of a step)' % (every, len(all_launches) // steps),
            'algorithmic_bytes_per_launch': float(np.mean(k_bytes))}
    roofline = spmm
    extras = {'spmm_launches_per_step': len(all_launches) // steps, 'spmm_ms_per_step': spmm_ms_step}
    # every position of a step priced on its own (cfg 3's launches range 76-144 us): what the launch is, the
    # bytes it has to move, its time and its fraction of the HBM roofline
    per_step = len(all_launches) // steps
    if per_step * steps == len(all_launches):
        pos_ms, pos_b = [[] for _ in range(per_step)], [[] for _ in range(per_step)]
        timed_ids = {id(r): (ms, b) for r, ms, b in zip(prof, k_ms, k_bytes)}
        desc = [None] * per_step
        for i, r in enumerate(all_launches):
            p_ = i % per_step
            if id(r) in timed_ids:
                pos_ms[p_].append(timed_ids[id(r)][0]); pos_b[p_].append(timed_ids[id(r)][1])
            m = _meta(r)
            desc[p_] = '%s%s%s%s%s' % ('A^T' if m['which'] == 'bwd' else 'A', ' x%d views' % m['views'] if m['views'] > 1 else '',
                                       ' +perturbation(%s)' % ('philox' if m['philox'] else 'table') if m['perturbed'] else '',
                                       ' +acc' if r[4] else '', '' if r[5] else ' (no Y)')
        spmm['launch_by_position_in_step'] = [
            {'launch': desc[p_], 'us': round(float(np.mean(pos_ms[p_])) * 1e3, 2) if pos_ms[p_] else None,
             'algorithmic_MB': round(float(np.mean(pos_b[p_])) / 1e6, 1) if pos_b[p_] else None,
             'frac': round(float(np.mean(pos_b[p_])) / (float(np.mean(pos_ms[p_])) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pos_ms[p_] else None,
             'samples': len(pos_ms[p_])} for p_ in range(per_step)]
    if inf:
        i_ms = [a.elapsed_time(b) for a, b, *_ in inf]
        flops = [ops.infonce_issued_flops(r[2], r[3], r[4], r[5], r[6]) for r in inf]
        unit = flops[0][1]
        peak = MFMA_BF16_PEAK_TF if unit in ('bf16', 'fp16') else MFMA_F32_PEAK_TF      # (dense fp16 and bf16 MFMA peaks are equal: 2.5 PFLOP/s)
        ach = float(np.sum([f for f, _ in flops])) / (float(np.sum(i_ms)) * 1e-3) / 1e12
        eq = float(np.sum([8.0 * r[3] * r[4] * r[5] / 2 for r in inf]))      # fp32-equivalent 8 B M d per forward + backward pair
        pairs = float(np.sum([r[3] * r[4] for r in inf if r[2] == 'fwd']))
        inf_ms_step = float(np.sum(i_ms)) / steps
        infonce = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                   'kernel': 'infonce_bwd_lds_kernel (anchor-gradient role + row sums in the forward call, all-gradient role in the backward call; '
                             '%s MFMA)' % ({'bf16': 'three bf16 planes / six v_mfma_f32_32x32x16_bf16 terms per product',
                                            'fp16': 'h3, the default: two fp16 planes / three v_mfma_f32_32x32x16_f16 terms per product'}.get(unit, 'v_mfma_f32_32x32x2_f32')),
                   'datatype': unit,
                   'issued_flops_per_step': float(np.sum([f for f, _ in flops])) / steps,
                   'fp32_equivalent_flops_per_step': eq / steps, 'fp32_equivalent_TFLOPs': eq / (float(np.sum(i_ms)) * 1e-3) / 1e12,
                   'calls_per_step': len(inf) // steps, 'ms_per_step': inf_ms_step, 'pairs_per_s': pairs / (float(np.sum(i_ms)) * 1e-3),
                   'call_timing': 'HIP events around every fused-InfoNCE forward and backward call of the timed region (preparation and '
                                  'finishing launches of a call included)'}
        extras['spmm_roofline'] = spmm
        if inf_ms_step > spmm_ms_step:
            roofline = infonce
        else:
            extras['infonce_roofline'] = infonce
    line = {'metric': 'propagation_edges_per_sec', 'value': edges * steps / elapsed, 'unit': 'edges/s', 'n_gpus': 1, 'steps': steps,
            'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            # the arithmetic of the line's DOMINANT kernel: the SpMM / BPR arithmetic is fp32; a step dominated by the fused InfoNCE runs its
            # B x M products on two fp16 planes with fp32 accumulation (h3) -- the exact-fp32 step time is extras.ms_per_step_infonce_fp32
            'dtype': 'f32' if roofline is spmm else 'f32 tables and accumulation; InfoNCE products on 2 fp16 planes x 3 MFMA terms (h3, 22-bit operands)',
            'data': 'real yelp interactions' if graph_name == 'yelp-real' else 'synthetic',
            'config': {'workload': '%s: %s cal_loss+backward, %dx%d, %d interactions (%d directed entries), d=%d, L=%d, B=%d, augmentation '
                                   'randomness computed in the kernels (model.device_rng)' % (tag, desc, trn.shape[0], trn.shape[1], trn.nnz, 2 * trn.nnz, d, L, B),
                       'edges_per_step': edges, 'parallelism': 'single GPU'},
            'roofline': roofline}
    extras['ms_per_step_as_one_hip_graph'] = graph_ms
    if graph_err:
        extras['hip_graph_error'] = graph_err
    # Both forms under fixed keys, the headline chosen by a FIXED rule per config (not by a threshold on the measurement): cfg 1's step is
    # ~20 launches of a few microseconds under 0.4 ms of Python, so its line is the captured step a training run executes with
    # train.hip_graph; cfg 3 / cfg 4 are quoted on eager launches.  The step is cal_loss + backward: optimizer.step() is NOT in it.
    line['ms_per_step_eager'], line['value_eager'] = line['ms_per_step'], line['value']
    line['ms_per_step_graph'] = graph_ms
    line['value_graph'] = None if graph_ms is None else edges / (graph_ms * 1e-3)
    line['headline_form'] = HEADLINE_FORM[tag] if graph_ms is not None else 'eager'
    line['step_excludes'] = 'optimizer.step()'
    if line['headline_form'] == 'graph':
        extras['ms_per_step_eager_launches'] = line['ms_per_step']
        line['ms_per_step'] = graph_ms
        line['value'] = line['value_graph']
        line['config']['workload'] += '; step replayed as one captured hipGraph'
    if inf:      # the same eager step with the InfoNCE products in the other arithmetics (the line's `dtype` is f32: `fp32` is the exact-fp32 MFMA,
        # h3 / x6 carry fp32-level error in fp16 / bf16 planes -- see roofline_infonce of the headline line)
        for prec in ('fp32', 'x6'):
            try:
                model.infonce_precision = prec
                extras['ms_per_step_infonce_' + prec] = _timed(step, max(5, steps // 3), 2)
            except Exception as exc:
                extras['ms_per_step_infonce_%s_error' % prec] = repr(exc)[:200]
        model.infonce_precision = None
    del model, dh
    torch.cuda.empty_cache()
    if with_parity_mode:      # the reference's own CPU generator stream, continued on the device (bit-identical masks / noise)
        try:
            trn_p, dh_p, model_p, _ = _build(tag, dev, device_rng=False)

            def step_p():
                model_p.zero_grad(set_to_none=True)
                loss, _ = model_p.cal_loss(batch)
                loss.backward()
            rng_mod.enable_host_replay(dev)
            try:
                extras['ms_per_step_parity_mode_generator_on_device'] = _timed(step_p, max(5, steps // 3), 2)
            finally:
                rng_mod.disable_host_replay()
            del model_p, dh_p
            torch.cuda.empty_cache()
        except Exception as exc:
            extras['parity_mode_error'] = repr(exc)[:300]
    if with_cpu:
        try:
            best, n, thr = _cpu_step_baseline(tag, trn, mcfg, batch, cpu_budget_s)
            line['cpu_baseline'] = {'value': edges / best, 'unit': 'edges/s', 'cores': thr, 'kind': 'port',
                                    'ms_per_step': best * 1e3,
                                    'sample': '%d step(s) of the oracle restatement of the reference %s step (cal_loss + backward, '
                                              'oracle/ref_expr.py) on the same graph and batch, thread counts 16 / 64 / all %d cores tried while '
                                              'the %.0f s budget lasted: fastest %.1f ms with %d threads' % (n, model_name, os.cpu_count(), cpu_budget_s, best * 1e3, thr)}
        except Exception as exc:
            line['cpu_baseline'] = {'error': repr(exc)[:300]}
    line['extras'] = extras
    return line


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--configs', default='cfg1,cfg3,cfg4')
    args = ap.parse_args()
    for tag in args.configs.split(','):
        print(json.dumps(run_config(tag, args.steps)), flush=True)
