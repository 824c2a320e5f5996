#!/usr/bin/env python
"""Benchmark of the general-CF hot path on MI355X (contract: see the task brief).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the LightGCN hot path over one batch: `cal_loss` + `backward`
(reference lightgcn.py:45-56) -- forward L-layer propagation (fused CSR-SpMM launches with the
layer sum), fused BPR over B=4096 triples, L2 regularizer, and the whole backward pass -- on the
amazon-book-shaped graph of BASELINE.json configs[1] (52,643 x 91,599, 2,380,730 train
interactions, d=64, L=3, keep_rate 1.0): 2*L SpMM launches, 2*L*nnz propagated directed edges.  Inputs (CSR, embeddings) are resident in HBM before the timed
region.  value = directed edges propagated per second, whole job.

At N>1 the embedding rows are dealt cyclically over the ranks (sslrec_amd/shard.py): one
RCCL all-gather + one local SpMM per layer, forward and backward (--shard-mode pipelined: the exchange as one
broadcast per source rank overlapped with per-source block products); the same graph is used at every N (strong
scaling).  `multi_gpu` reports local SpMM time, the step's collectives timed alone, and the overlap fraction.

The JSON line also carries
  roofline     : HBM roofline of the dominant kernel (the SpMM), from HIP-event timings of
                 every SpMM launch inside the timed region and the algorithmic bytes of
                 SURVEY.md §8(d) (entries*8 + row segments*8 + streams*16 + X read once + Y written once
                 [+ 2 passes for the fused accumulator]);
  cpu_baseline : the reference's CPU expression (torch.spmm over the uncoalesced COO,
                 lightgcn.py:28-29) timed on this box's cores on the SAME graph (oracle port);
  extras       : masked-propagation rate, fused InfoNCE pairs/s with its FP32-MFMA roofline
                 fraction, and full training-step times (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
MFMA_F32_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32 dense peak


def build_graph_host(name):
    """interaction matrix + normalized bipartite adjacency as numpy (host logic of the data handler)"""
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
    load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': name}})
    dh = DataHandlerGeneralCF()
    trn = dh._load_one_mat(dh.trn_file)
    configs['data']['user_num'], configs['data']['item_num'] = trn.shape
    adj = dh._make_torch_adj(trn)
    idx = adj._indices().numpy()
    return trn, idx[0], idx[1], adj._values().numpy(), trn.shape[0] + trn.shape[1]


def xavier_tables(n_user, n_item, d, seed=2023):
    torch.manual_seed(seed)
    ue = torch.nn.init.xavier_uniform_(torch.empty(n_user, d))
    ie = torch.nn.init.xavier_uniform_(torch.empty(n_item, d))
    return ue, ie


def cpu_baseline(rows, cols, vals, n, d, budget_s=15.0):
    """oracle port of the reference path on the host cores, bounded sample"""
    from oracle import ref_expr as R
    torch.set_num_threads(os.cpu_count())
    adj = R.torch_adj_from(np.vstack([rows, cols]), vals, n)
    x = torch.randn(n, d)
    R.propagate(adj, x)                                        # warm-up
    t0 = time.perf_counter()
    reps, times = 0, []
    while reps < 10 and (time.perf_counter() - t0) < budget_s:
        t1 = time.perf_counter()
        R.propagate(adj, x)
        times.append(time.perf_counter() - t1)
        reps += 1
    med = float(np.median(times))
    return {'value': vals.size / med, 'unit': 'edges/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d x torch.spmm(uncoalesced COO %dx%d nnz=%d, X[%d,%d]) forward, median %.1f ms'
                      % (reps, n, n, vals.size, n, d, med * 1e3)}


def time_events(fn, reps, warmup=2):
    for _ in range(warmup):
        fn()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))   # ms


def extras_single_gpu(trn, rows, cols, vals, n, graph, d, L, dev):
    """secondary figures (not the headline): masked SpMM, fused InfoNCE, full model steps"""
    from sslrec_amd import ops
    from sslrec_amd.graph import DroppedView
    out = {}
    n_user, n_item = trn.shape
    x = torch.randn(n, d, device=dev)
    # plain SpMM, the figure SURVEY §8d quotes
    ms = time_events(lambda: ops.spmm_raw(graph, x, 'fwd'), 20)
    out['spmm_plain_us'] = ms * 1e3
    out['spmm_plain_edges_per_s'] = graph.nnz / (ms * 1e-3)
    lay = graph.fwd.swept(d)
    out['spmm_plain_kernel'] = 'swept (LDS accumulators)' if lay is not None else 'streamed'
    out['spmm_plain_hbm_frac'] = (lay or graph.fwd).algorithmic_bytes(d) / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
    if lay is not None:       # the row-streamed kernel on the same graph, for comparison
        os.environ['SSLREC_SPMM_SWEPT'] = '0'
        try:
            from sslrec_amd.graph import PropGraph
            g_str = PropGraph(rows, cols, vals, (n, n), dev)
            ms_s = time_events(lambda: ops.spmm_raw(g_str, x, 'fwd'), 20)
            out['spmm_streamed_kernel_us'] = ms_s * 1e3
            del g_str
        finally:
            os.environ.pop('SSLREC_SPMM_SWEPT')
    out['spmm_gather_model_GBs'] = (graph.nnz * (8 + 4 * d) + n * d * 4) / (ms * 1e-3) / 1e9
    # stock comparator on the SAME GPU: what the reference executes there -- torch.spmm over the
    # uncoalesced COO (PyTorch re-coalesces and calls hipSPARSE on every call), lightgcn.py:28-29
    try:
        coo = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([rows, cols])), torch.from_numpy(vals), (n, n),
                                      check_invariants=False).to(dev)
        ms_t = time_events(lambda: torch.spmm(coo, x), 5, warmup=1)
        out['stock_torch_spmm_uncoalesced_coo_us'] = ms_t * 1e3
        csr = coo.coalesce().to_sparse_csr()
        ms_c = time_events(lambda: csr @ x, 10, warmup=2)
        out['stock_torch_spmm_csr_us'] = ms_c * 1e3
        out['speedup_vs_stock_reference_path'] = ms_t / ms
        del coo, csr
    except Exception as exc:
        out['stock_torch_spmm_error'] = repr(exc)
    # edge-dropped (keep 0.5) view: compaction once + SpMM on kept edges
    keep = (torch.rand(graph.nnz) + 0.5).floor().bool()
    prep = (lambda v: v.masked('fwd', d)) if graph.fwd.swept(d) is not None else (lambda v: v.compact('fwd', d))
    t_c = time_events(lambda: prep(DroppedView(graph, keep)), 5, warmup=1)
    view = DroppedView(graph, keep)
    prep(view)
    ms_m = time_events(lambda: ops.spmm_raw(view, x, 'fwd'), 20)
    out['masked_keep0.5_spmm_us'] = ms_m * 1e3
    out['masked_kept_edges_per_s'] = view.n_kept() / (ms_m * 1e-3)
    out['edge_drop_prepare_us_incl_mask_h2d'] = t_c * 1e3        # swept: mask in place; streamed: per-row compaction
    # fused InfoNCE, SimGCL item term of cfg 3: B=4096 anchors vs all 91,599 items
    B, temp = 4096, 0.2
    t1 = (torch.randn(n_item, d, device=dev) * 0.1).requires_grad_(True)
    t2 = (torch.randn(n_item, d, device=dev) * 0.1).requires_grad_(True)
    idx = torch.randint(0, n_item, (B,), device=dev)
    ms_f = time_events(lambda: ops.infonce_loss_gathered(t1.detach(), t2.detach(), idx, temp), 10)

    def fb():
        t1.grad = t2.grad = None
        ops.infonce_loss_gathered(t1, t2, idx, temp).backward()
    ms_fb = time_events(fb, 10)
    pairs = B * n_item
    try:   # stock comparator: the reference's materializing expression (loss_utils.py:30-39) in PyTorch on this GPU
        def stock():
            t1.grad = t2.grad = None
            e1, e2 = t1[idx], t2[idx]
            n1 = e1 / torch.sqrt(1e-8 + e1.square().sum(-1, keepdim=True))
            n2 = e2 / torch.sqrt(1e-8 + e2.square().sum(-1, keepdim=True))
            na = t2 / torch.sqrt(1e-8 + t2.square().sum(-1, keepdim=True))
            loss = (-(n1 * n2 / temp).sum(-1) + torch.log(torch.exp(n1 @ na.T / temp).sum(-1))).sum()
            loss.backward()
        out['stock_torch_infonce_fwdbwd_ms'] = time_events(stock, 3, warmup=1)
        torch.cuda.empty_cache()
    except Exception as exc:
        out['stock_torch_infonce_error'] = repr(exc)
    out['infonce_precision'] = os.environ.get('SSLREC_INFONCE_PRECISION', 'x6') + ' (x6 = 3 bf16 planes / 6 MFMA terms, fp32-level error)'
    out['infonce_fwd_ms'] = ms_f
    out['infonce_fwd_pairs_per_s'] = pairs / (ms_f * 1e-3)
    out['infonce_fwdbwd_ms'] = ms_fb
    out['infonce_fwdbwd_pairs_per_s'] = pairs / (ms_fb * 1e-3)
    # the other precisions of the same call: exact-fp32 MFMA (with its fraction of the 157.3 TFLOP/s FP32-MFMA peak: 2BMd
    # flops forward, 8BMd forward+backward with the recomputation) and the opt-in fast modes
    saved = os.environ.get('SSLREC_INFONCE_PRECISION')
    try:
        for prec in ('fp32', 'x36', 'x3'):
            os.environ['SSLREC_INFONCE_PRECISION'] = prec
            f_ms = time_events(lambda: ops.infonce_loss_gathered(t1.detach(), t2.detach(), idx, temp), 10)
            fb_ms = time_events(fb, 10)
            out['infonce_%s_fwd_ms' % prec], out['infonce_%s_fwdbwd_ms' % prec] = f_ms, fb_ms
            if prec == 'fp32':
                out['infonce_fp32_fwd_mfma_frac'] = 2.0 * pairs * d / (f_ms * 1e-3) / (MFMA_F32_PEAK_TF * 1e12)
                out['infonce_fp32_fwdbwd_mfma_frac'] = 8.0 * pairs * d / (fb_ms * 1e-3) / (MFMA_F32_PEAK_TF * 1e12)
    finally:
        if saved is None:
            os.environ.pop('SSLREC_INFONCE_PRECISION', None)
        else:
            os.environ['SSLREC_INFONCE_PRECISION'] = saved
    # all-rank evaluation (fused MFMA + train-CSR mask + top-k) and the device negative sampler
    try:
        import scipy.sparse as sp
        from sslrec_amd.rng import PhiloxState
        csr = sp.csr_matrix(trn)
        csr.sort_indices()
        trn_csr = (torch.from_numpy(csr.indptr.astype(np.int64)).to(dev), torch.from_numpy(csr.indices.astype(np.int64)).to(dev))
        ue, ie = torch.randn(n_user, d, device=dev) * 0.1, torch.randn(n_item, d, device=dev) * 0.1
        users_all = torch.arange(n_user, device=dev)
        out['eval_topk40_all_%d_users_ms' % n_user] = time_events(lambda: ops.eval_topk(ue, ie, users_all, 40, trn_csr), 5, 1)
        out['eval_topk40_1024_users_ms'] = time_events(lambda: ops.eval_topk(ue, ie, users_all[:1024], 40, trn_csr), 10, 2)

        def stock_eval():      # the reference's expression on this GPU for ONE batch of 1024 users (dense mask assumed resident)
            sc = ue[:1024] @ ie.T
            return torch.topk(sc * (1 - mask1024) - 1e8 * mask1024, 40)[1]
        mask1024 = torch.from_numpy(csr[:1024].toarray().astype(np.float32)).to(dev)
        out['stock_torch_eval_1024_users_ms_mask_resident'] = time_events(stock_eval, 5, 1)
        del mask1024
        coo = trn.tocoo()
        inter_users = torch.from_numpy(coo.row.astype(np.int64)).to(dev)
        state = PhiloxState(dev, seed=1)
        state.advance()
        out['sample_negs_%d_interactions_ms' % inter_users.numel()] = time_events(lambda: ops.sample_negs(inter_users, trn_csr, n_item, state, stream_id=1), 10, 2)
    except Exception as exc:
        out['eval_sampler_error'] = repr(exc)
    # full training steps through the model classes (cal_loss + backward), parity-mode RNG on the CPU
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
    from sslrec_amd.models.bulid_model import build_model
    # three ways of drawing the augmentation randomness: the reference's CPU draws copied over (parity), the same numbers
    # produced by the CPU generator's algorithm on the device (parity too: sslrec_amd.rng.HostGeneratorReplay, what the
    # Trainer uses by default), Philox inside the kernels (model.device_rng)
    from sslrec_amd import rng as rng_mod
    for model_name in ('lightgcn', 'simgcl'):
        for mode in ('cpu_rng_parity', 'parity_generator_on_device', 'device_rng'):
            load_config(model_name, device=dev, overrides={'data': {'synthetic': 'amazon-book'},
                                                           'model': {'embedding_size': d, 'layer_num': L,
                                                                     'device_rng': mode == 'device_rng'}})
            dh = DataHandlerGeneralCF()
            dh.trn_mat = trn
            configs['data']['user_num'], configs['data']['item_num'] = trn.shape
            dh.torch_adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([rows, cols])), torch.from_numpy(vals),
                                                   (n, n), check_invariants=False).to(dev)
            dh.torch_adj._sslrec_graph = graph
            model = build_model(dh).to(dev)
            batch = [torch.randint(0, n_user, (B,), device=dev), torch.randint(0, n_item, (B,), device=dev),
                     torch.randint(0, n_item, (B,), device=dev)]

            def step():
                model.zero_grad(set_to_none=True)
                loss, _ = model.cal_loss(batch)
                loss.backward()
            if mode == 'parity_generator_on_device':
                rng_mod.enable_host_replay(dev)
            try:
                out['%s_step_ms_%s' % (model_name, mode)] = time_events(step, 5 if mode == 'cpu_rng_parity' else 10, 1)
            finally:
                rng_mod.disable_host_replay()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='amazon-book')
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    # N > 1: 'feature' (default) = every GPU holds all rows and d / N columns, no collective in the propagation
    # (sslrec_amd/feature_shard.py); the others = row-sharded tables with one exchange per layer (sslrec_amd/shard.py)
    ap.add_argument('--shard-mode', default='feature', choices=['feature', 'all_gather', 'pipelined', 'reduce_scatter'])
    ap.add_argument('--eager-step', action='store_true',
                    help='feature mode: issue the step as ~40 eager launches instead of two captured hipGraphs around the all-gather')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        sys.exit('bench.py --gpus %d must be launched with %d processes (torch.distributed.run); WORLD_SIZE=%d'
                 % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (no CPU fallback)')
    # SSLREC_BENCH_ONE_DEVICE=1: every rank on cuda:0 with gloo (host-staged) collectives -- exercises the N > 1 code path
    # on a single-GPU box; the numbers of such a run mean nothing
    one_device = os.environ.get('SSLREC_BENCH_ONE_DEVICE') == '1'
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    import torch.distributed as dist
    if world > 1:
        if one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(dev))

    from sslrec_amd import ops
    d, L = args.dim, args.layers
    trn, rows, cols, vals, n = build_graph_host(args.workload)
    ue, ie = xavier_tables(trn.shape[0], trn.shape[1], d)
    e0_full = torch.cat([ue, ie])
    g_full = torch.randn(n, d, generator=torch.Generator().manual_seed(7)) * 1e-3

    B = 4096                                                   # train.batch_size of every target yml
    bgen = torch.Generator().manual_seed(11)
    batch = [torch.randint(0, trn.shape[0], (B,), generator=bgen), torch.randint(0, trn.shape[1], (B,), generator=bgen),
             torch.randint(0, trn.shape[1], (B,), generator=bgen)]
    reg_weight = 1.0e-8
    graphed = False
    if world == 1:
        from sslrec_amd.graph import PropGraph
        graph = PropGraph(rows, cols, vals, (n, n), dev)
        e0 = e0_full.to(dev).requires_grad_(True)
        batch = [b.to(dev) for b in batch]

        def step():     # LightGCN cal_loss + backward (lightgcn.py:45-56) on the fused path, keep_rate 1.0
            e0.grad = None
            s = ops.propagate_sum(graph, e0, L)
            loss = ops.bpr_loss_stacked(s, trn.shape[0], *batch, divisor=B) + ops.sum_squares(e0, reg_weight)
            loss.backward()
    elif args.shard_mode == 'feature' and d % world == 0 and d // world in (8, 16, 32, 64, 128, 256):
        from sslrec_amd.feature_shard import FeatureSlicedGraphCF
        from sslrec_amd.graph import PropGraph
        graph = PropGraph(rows, cols, vals, (n, n), dev)
        model = FeatureSlicedGraphCF(graph, trn.shape[0], trn.shape[1], e0_full, L, world, rank)
        batch = [b.to(dev) for b in batch]

        def eager_step():     # same step on feature-sliced tables: local propagation of d / N columns, batch rows by one all-gather
            model.local_embeds.grad = None
            model.lightgcn_loss(batch, reg_weight).backward()
        step = eager_step
        if not args.eager_step:      # two hipGraph replays around the one collective: a GPU's share of the work is smaller than
            from sslrec_amd.feature_shard import GraphedLightGCNStep          # the host time of the eager launches
            try:
                gstep = GraphedLightGCNStep(model, B, reg_weight)
                captured = 1
            except Exception as exc:                          # capture refused on this box: every rank falls back to eager launches
                sys.stderr.write('rank %d: hipGraph capture failed (%r); eager step\n' % (rank, exc))
                captured = 0
            flag = torch.tensor([captured], dtype=torch.int32, device='cpu' if one_device else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            graphed = bool(flag.item())
            if graphed:
                def step():
                    gstep.step(batch)
    else:
        if args.shard_mode == 'feature':
            args.shard_mode = 'all_gather'          # d / N is not a width of the kernel: row shards
        from sslrec_amd.shard import ShardedGraph, ShardedGraphCF
        sg = ShardedGraph(rows, cols, vals, n, world, rank, dev)
        model = ShardedGraphCF(sg, trn.shape[0], trn.shape[1], e0_full, L, mode=args.shard_mode)
        batch = [b.to(dev) for b in batch]

        def step():     # same step on row-sharded tables: this rank's batch slice, collectives through autograd
            model.local_embeds.grad = None
            model.lightgcn_loss(batch, reg_weight).backward()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.PROFILE = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    launch_timing = 'HIP events around every SpMM launch of the timed region'
    if graphed:      # graph replays carry no per-launch events: the same step issued eagerly, right after the timed region
        ops.PROFILE = []
        for _ in range(5):
            eager_step()
        barrier()
        prof, ops.PROFILE = ops.PROFILE, None
        launch_timing = 'HIP events around every SpMM launch of 5 eager steps issued after the timed region (the timed steps are hipGraph replays)'
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if one_device else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    edges_per_step = 2 * L * int(vals.size)
    value = edges_per_step * args.steps / elapsed

    # roofline of the dominant kernel from the HIP-event timings of the timed region (this rank)
    k_ms = [a.elapsed_time(b) for a, b, *_ in prof]
    k_bytes = [plan.algorithmic_bytes(dd, acc=has_acc, write_y=want_y) for _, _, plan, dd, has_acc, want_y, *_ in prof]
    avg_s = float(np.mean(k_ms)) * 1e-3
    achieved = float(np.mean(k_bytes)) / avg_s / 1e9
    traffic = traffic_src = None
    tf = os.path.join(ROOT, 'profiles', 'spmm_traffic.json')
    if os.path.exists(tf) and world == 1:       # (the stamp belongs to the single-GPU kernel) PMC passes are separate runs (rocprofv3 --pmc): the committed file of the last profiled commit
        tj = json.load(open(tf))
        traffic = tj.get('hbm_bytes_per_launch')
        traffic_src = 'profiles/spmm_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of this command, round %s, commit %s' % (
            tj.get('measured_in_round'), tj.get('measured_at_commit'))
    roofline = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                'kernel': ('spmm_swept_kernel<%d> (LDS accumulators, column-swept)' % prof[0][3]) if type(prof[0][2]).__name__ == 'SweptLayout'
                          else 'spmm_stream_kernel<%d> (+long-row reduce)' % prof[0][3],
                'avg_launch_us': avg_s * 1e6, 'launches': len(k_ms), 'launch_timing': launch_timing,
                'algorithmic_bytes_per_launch': float(np.mean(k_bytes))}

    # multi-GPU: the same figure with the collective excluded (local SpMM launches only, this rank's
    # HIP-event timings) -- SURVEY.md §8e asks for edges/s with and without the per-layer collective
    multi = None
    if world > 1:
        from sslrec_amd.shard import all_gather_rows, all_reduce_sum, reduce_scatter_rows, rows_per_rank, shards_pipelined
        local_s = float(np.sum(k_ms)) * 1e-3 / (5 if graphed else args.steps)
        # the step's collectives ALONE (same sizes, same count: L forward + L-1 backward exchanges of [n_per, d] rows,
        # one [3B, d] all-reduce), timed without any compute between them
        n_per = rows_per_rank(n, world)
        xs = torch.randn(n_per, d, device=dev)
        small = torch.zeros(3 * B, d, device=dev)

        def exchanges():
            if args.shard_mode == 'feature':          # the step's only collective: the [3B, d/N] slices of the batch rows
                all_gather_rows(small[:, :d // world].contiguous(), world)
                return
            for _ in range(2 * L - 1):
                if args.shard_mode == 'pipelined':
                    for _q, _x in shards_pipelined(xs, world, rank):
                        pass
                elif args.shard_mode == 'reduce_scatter':
                    reduce_scatter_rows(torch.empty(n_per * world, d, device=dev), world)
                else:
                    all_gather_rows(xs, world)
            all_reduce_sum(small)
        for _ in range(3):
            exchanges()
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            exchanges()
        barrier()
        coll_s = (time.perf_counter() - t1) / 10
        step_s = elapsed / args.steps
        multi = {'local_spmm_ms': local_s * 1e3, 'collective_ms': coll_s * 1e3,
                 # share of the collective time that the step hides behind compute (0 when they simply add up; the
                 # step's other kernels -- BPR, regularizer -- make this a lower bound)
                 'overlap_frac': float(min(1.0, max(0.0, (local_s + coll_s - step_s) / coll_s))) if coll_s > 0 else None,
                 'edges_per_s_excluding_collective': edges_per_step / local_s,
                 'collective': args.shard_mode,
                 'collective_bytes_per_rank_per_layer': 0 if args.shard_mode == 'feature' else int(n * d * 4 * (world - 1) / world),
                 'collective_bytes_per_rank_per_step': int(3 * B * d * 4 * (world - 1) / world) if args.shard_mode == 'feature'
                 else int((2 * L - 1) * n * d * 4 * (world - 1) / world + 3 * B * d * 4)}
    if rank == 0:
        line = {
            'metric': 'propagation_edges_per_sec', 'value': value, 'unit': 'edges/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'LightGCN cal_loss+backward (propagation fwd+bwd, fused BPR, B=4096) on %s-shaped synthetic '
                                   'graph (%dx%d, E=%d, nnz=%d), d=%d, L=%d, keep_rate=1.0' % (args.workload, trn.shape[0], trn.shape[1], trn.nnz,
                                                                   vals.size, d, L),
                       'edges_per_step': edges_per_step,
                       'parallelism': 'single GPU' if world == 1 else
                       ('feature-sliced: all rows x %d of %d embedding columns per GPU, whole adjacency on each of %d GPUs, no collective in '
                        'the propagation, one [3B, d/N] all-gather per step%s' % (d // world, d, world, '; step = two hipGraph replays around the all-gather' if graphed else '')) if args.shard_mode == 'feature' else
                       'rows dealt cyclically over %d GPUs, one %s per layer' % (world, args.shard_mode)},
            'roofline': roofline,
        }
        if multi is not None:
            line['multi_gpu'] = multi
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(rows, cols, vals, n, d)
        if world == 1 and not args.no_extras:
            try:
                line['extras'] = extras_single_gpu(trn, rows, cols, vals, n, graph, d, L, dev)
            except Exception as exc:                      # extras never invalidate the headline
                line['extras'] = {'error': repr(exc)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
