#!/usr/bin/env python
"""Benchmark of the general-CF hot path on MI355X (contract: see the task brief).

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: bench.py starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the LightGCN hot path over one batch: `cal_loss` + `backward`
(reference lightgcn.py:45-56) -- forward L-layer propagation (fused CSR-SpMM launches with the
layer sum), fused BPR over B=4096 triples, L2 regularizer, and the whole backward pass -- on the
amazon-book-shaped graph of BASELINE.json configs[1] (52,643 x 91,599, 2,380,730 train
interactions, d=64, L=3, keep_rate 1.0): 2*L SpMM launches, 2*L*nnz propagated directed edges.  Inputs (CSR, embeddings) are resident in HBM before the timed
region.  value = directed edges propagated per second, whole job.

At N>1 two decompositions are timed in the same run (same graph at every N: strong scaling).  Headline (`value`): every GPU
holds all rows and d/N embedding columns plus the whole adjacency -- no collective in the propagation, one [3B, d/N]
all-gather per step (sslrec_amd/feature_shard.py).  Under `multi_gpu.row_sharded`: the partition BASELINE.json words --
the rows dealt cyclically over the ranks, one RCCL all-gather + one local SpMM per layer, forward and backward
(sslrec_amd/shard.py; --shard-mode all_gather | pipelined | reduce_scatter makes one of these the headline instead).
Each carries local SpMM time, the step's collectives timed alone, and the overlap fraction.

The JSON line also carries
  roofline     : HBM roofline of the dominant kernel (the SpMM), from HIP-event timings of
                 every 5th SpMM launch inside the timed region (rotating through the launches of a step; see run_eager)
                 and the algorithmic bytes of SURVEY.md §8(d) (entries*8 + (rows+1)*4 + X read once + Y written once
                 [+ 2 passes for the fused accumulator]);
  cpu_baseline : the reference's CPU expression (torch.spmm over the uncoalesced COO,
                 lightgcn.py:28-29) timed on this box's cores on the SAME graph (oracle port);
  extras       : masked-propagation rate, fused InfoNCE pairs/s with its FP32-MFMA roofline
                 fraction, and full training-step times (N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TF = 2500.0        # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline includes 2:1 sparsity)
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
    """oracle port of the reference path on the host cores, bounded sample.  torch's CPU sparse product does not scale to hundreds
    of threads (the GPU box has 256 cores), so 16 / 64 / all cores are tried while the budget lasts and the best is reported with
    the thread count it used"""
    from oracle import ref_expr as R
    adj = R.torch_adj_from(np.vstack([rows, cols]), vals, n)
    x = torch.randn(n, d)
    best = None
    t_all = time.perf_counter()
    for thr in sorted({min(16, os.cpu_count()), min(64, os.cpu_count()), os.cpu_count()}):
        if best is not None and time.perf_counter() - t_all > 0.7 * budget_s:
            break
        torch.set_num_threads(thr)
        R.propagate(adj, x)                                        # warm-up
        times = []
        while len(times) < 5 and (time.perf_counter() - t_all) < budget_s:
            t1 = time.perf_counter()
            R.propagate(adj, x)
            times.append(time.perf_counter() - t1)
        if times and (best is None or float(np.median(times)) < best[0]):
            best = (float(np.median(times)), thr, len(times))
    torch.set_num_threads(os.cpu_count())
    med, thr, reps = best
    return {'value': vals.size / med, 'unit': 'edges/s', 'cores': thr, 'kind': 'port',
            'sample': '%d x torch.spmm(uncoalesced COO %dx%d nnz=%d, X[%d,%d]) forward, median %.1f ms with %d threads (best of 16 / 64 / all %d cores)'
                      % (reps, n, n, vals.size, n, d, med * 1e3, thr, os.cpu_count())}


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


def gather_ceiling(dev, n_user, n_item, d, nnz, algorithmic_bytes, avg_launch_s):
    """The rate at which this GPU gathers random rows of d floats, measured in this process beside the SpMM it bounds
    (sslrec_debug_gather_rows = tools/micro/gather_ceiling.hip inside the library): a kernel that fetches one neighbour row per entry
    cannot finish a launch faster than nnz * 4d bytes / that rate, whatever its layout.  Three tables: one that fits every L2 (2,048
    rows), the two row classes of the bipartite adjacency as the swept kernel sees them (an XCD's rows reference ONE class), and the
    whole stacked table."""
    import ctypes as C
    from sslrec_amd import _lib
    lib = _lib.load()
    row_bytes = 4 * d
    scratch = torch.zeros(4096, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def rate(n_rows):
        x = torch.randn(n_rows * d, device=dev)
        cnt = C.c_int64(0)

        def run():
            _lib.check(lib.sslrec_debug_gather_rows(x.data_ptr(), n_rows, row_bytes, 64, 256, scratch.data_ptr(), C.addressof(cnt), st), 'sslrec_debug_gather_rows')
        ms = time_events(run, 10)
        return cnt.value * row_bytes / (ms * 1e-3) / 1e12      # TB/s
    out = {'row_bytes': row_bytes, 'kernel': 'sslrec_debug_gather_rows: 256 workgroups x 16 waves, 8 x 16-byte loads in flight per wave, random rows'}
    out['gather_TBps_L2_resident'] = rate(2048)
    out['gather_TBps_user_table_%d_rows' % n_user] = r_u = rate(n_user)
    out['gather_TBps_item_table_%d_rows' % n_item] = r_i = rate(n_item)
    out['gather_TBps_at_table_size'] = rate(n_user + n_item)
    gather_bytes = nnz * row_bytes
    # half of the entries gather user rows, half item rows (the adjacency is symmetric)
    floor_random_s = 0.5 * gather_bytes / (r_u * 1e12) + 0.5 * gather_bytes / (r_i * 1e12)
    floor_l2_s = gather_bytes / (out['gather_TBps_L2_resident'] * 1e12)
    out['gather_bytes_per_launch'] = gather_bytes
    out['launch_us_floor_if_every_gather_hit_L2'] = floor_l2_s * 1e6
    out['launch_us_floor_random_order_gathers'] = floor_random_s * 1e6
    out['frac_if_every_gather_hit_L2'] = algorithmic_bytes / floor_l2_s / 1e9 / HBM_PEAK_GBS
    out['frac_if_gathers_ran_at_table_size_rate'] = algorithmic_bytes / floor_random_s / 1e9 / HBM_PEAK_GBS
    frac = algorithmic_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS
    out['frac_of_ceiling'] = frac / out['frac_if_every_gather_hit_L2']
    out['note'] = ('`frac_if_every_gather_hit_L2` is what ANY one-gather-per-entry kernel could reach on this graph if all of its %d row '
                   'fetches (%.2f GB) came out of an L2 at the measured L2-resident rate; `frac_of_ceiling` = roofline.frac over it.  The column '
                   'sweep is why the kernel runs FASTER than random-order gathers of the same tables would (launch_us_floor_random_order_gathers).'
                   % (nnz, gather_bytes / 1e9))
    return out


def other_graphs_roofline(d, dev, l2_gather_TBps):
    """The same `roofline` block for two graphs WITH structure: the real yelp interactions (BASELINE cfg 4's
    data, 69,534 rows) and the amazon-book-shaped generator with 64 planted communities (p_in 0.95) -- both with the plan builder's
    automatic row -> XCD co-clustering (kept when it lowers the layout's distinct (XCD, column) pairs by > 25 %).  The headline graph
    (power-law degrees, no communities) offers the column sweep nothing to cluster; these do."""
    from sslrec_amd import ops
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils import synth
    from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
    from sslrec_amd.graph import PropGraph
    import scipy.sparse as sp
    out = {}
    u, i, e = synth.SHAPES['amazon-book']
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'yelp_lightgcn_d64_L2.npz'))
    U, I = (int(x_) for x_ in z['shape'])
    graphs = (('yelp-real', lambda: sp.coo_matrix((np.ones(z['trn_row'].size), (z['trn_row'], z['trn_col'])), shape=(U, I))),
              ('amazon-book-shape, 64 planted communities (p_in 0.95)', lambda: synth.community_bipartite(u, i, e, 64, 0.95)))
    for name, make in graphs:
        trn_ = sp.coo_matrix((make() != 0).astype(np.float32))          # the data handler's own host logic (_load_one_mat, _make_torch_adj)
        load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': 'tiny'}})
        dh_ = DataHandlerGeneralCF()
        configs['data']['user_num'], configs['data']['item_num'] = trn_.shape
        adj_ = dh_._make_torch_adj(trn_)
        idx, vals_, n_ = adj_._indices().numpy(), adj_._values().numpy(), trn_.shape[0] + trn_.shape[1]
        rec = {'n_rows': int(n_), 'nnz': int(vals_.size)}
        x = torch.randn(n_, d, device=dev)
        acc = torch.randn(n_, d, device=dev)
        for tag, env in (('auto', None), ('no_coclustering', '0')):
            if env is None:
                os.environ.pop('SSLREC_XCD_CLUSTER', None)
            else:
                os.environ['SSLREC_XCD_CLUSTER'] = env
            try:
                g = PropGraph(idx[0], idx[1], vals_, (n_, n_), dev)
                lay = g.fwd.swept(d)
                if lay is None:
                    rec[tag] = {'kernel': 'streamed (no swept layout)'}
                    continue
                y = torch.empty_like(x)
                ms_p = time_events(lambda: ops.spmm_raw(g, x, 'fwd', y=y), 20)
                ms_a = time_events(lambda: ops.spmm_raw(g, x, 'fwd', y=y, acc_in=acc, acc_out=acc), 20)
                bp, ba = lay.algorithmic_bytes(d), lay.algorithmic_bytes(d, acc=True)
                floor_s = vals_.size * 4 * d / (l2_gather_TBps * 1e12)
                rec[tag] = {'xcd_col_pairs': int(lay.xcd_col_pairs), 'fabric_read_floor_MB': lay.xcd_col_pairs * d * 4 / 1e6,
                            'plain': {'launch_us': ms_p * 1e3, 'algorithmic_bytes': bp, 'achieved_GBps': bp / (ms_p * 1e-3) / 1e9,
                                      'frac': bp / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBS},
                            'fused_accumulator': {'launch_us': ms_a * 1e3, 'algorithmic_bytes': ba, 'achieved_GBps': ba / (ms_a * 1e-3) / 1e9,
                                                  'frac': ba / (ms_a * 1e-3) / 1e9 / HBM_PEAK_GBS},
                            'frac_if_every_gather_hit_L2_plain': bp / floor_s / 1e9 / HBM_PEAK_GBS}
                del g
            finally:
                os.environ.pop('SSLREC_XCD_CLUSTER', None)
        a_, b_ = rec.get('auto', {}), rec.get('no_coclustering', {})
        if 'xcd_col_pairs' in a_ and 'xcd_col_pairs' in b_:
            rec['coclustering_kept_by_auto'] = a_['xcd_col_pairs'] < 0.75 * b_['xcd_col_pairs']
        out[name] = rec
        del x, acc
        torch.cuda.empty_cache()
    return out


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
    # the fused-accumulator launches of a propagation forward + backward by the DEVICE's wall clock (first workgroup start to last
    # workgroup end, sslrec_debug_stamp_next_launch) next to HIP events around the same launches: what lies between the two is
    # the kernel boundary (dispatch, end-of-kernel write-back of the dirty L2 lines, the event records themselves)
    try:
        e0_ = torch.randn(n, d, device=dev, requires_grad=True)
        gt_ = torch.randn(n, d, device=dev)

        def fb_():
            e0_.grad = None
            ops.propagate_sum(graph, e0_, L).backward(gt_)
        for _ in range(3):
            fb_()
        st = ops.StampLog(dev, 256)
        ops.STAMPS, ops.PROFILE = st, []
        for _ in range(10):
            fb_()
        torch.cuda.synchronize()
        prof_, ops.PROFILE, ops.STAMPS = ops.PROFILE, None, None
        out['spmm_fused_launch_us_by_device_clock'] = float(np.mean([ms_ for _, ms_, _ in st.read()])) * 1e3
        out['spmm_fused_launch_us_by_hip_events_same_launches'] = float(np.mean([a.elapsed_time(b) for a, b, *_ in prof_])) * 1e3
        del e0_, gt_
    except Exception as exc:
        ops.STAMPS = ops.PROFILE = None
        out['spmm_device_clock_error'] = repr(exc)
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
    out['infonce_precision'] = os.environ.get('SSLREC_INFONCE_PRECISION', 'h3') + ' (h3 = 2 fp16 planes / 3 MFMA terms, x6 = 3 bf16 planes / 6 terms: both fp32-level error)'
    out['infonce_fwd_ms'] = ms_f
    out['infonce_fwd_pairs_per_s'] = pairs / (ms_f * 1e-3)
    out['infonce_fwdbwd_ms'] = ms_fb
    out['infonce_fwdbwd_pairs_per_s'] = pairs / (ms_fb * 1e-3)
    # the other precisions of the same call: exact-fp32 MFMA (with its fraction of the 157.3 TFLOP/s FP32-MFMA peak: 2BMd
    # flops forward, 8BMd forward+backward with the recomputation) and the opt-in fast modes
    saved = os.environ.get('SSLREC_INFONCE_PRECISION')
    try:
        for prec in ('fp32', 'x6', 'x36', 'x3'):
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


def infonce_roofline(n_item, d, dev, B=4096, temp=0.2):
    """The fused InfoNCE of BASELINE cfg 3's item term (cal_infonce_loss, reference models/loss_utils.py:30-39, call site
    simgcl.py:49: B = 4096 anchors against ALL item rows of view 2) as a roofline block: bound = the matrix cores.  Per precision
    mode: the forward alone (no gradient wanted: the row-sum kernel), the differentiated forward (it also accumulates the
    anchor-gradient sums, SSLREC_INFONCE_FWD_W), forward + backward; issued matrix flops (x6: six bf16 MFMA terms per product and the
    B x M score tiles formed twice per forward + backward; fp32: v_mfma_f32_32x32x2_f32) against the dense peak of that datatype, and
    the fp32-equivalent 8 B M d of SURVEY.md 8d beside it.  Timed with HIP events on the launch stream, median of 10."""
    from sslrec_amd import ops
    t1 = (torch.randn(n_item, d, device=dev) * 0.1).requires_grad_(True)
    t2 = (torch.randn(n_item, d, device=dev) * 0.1).requires_grad_(True)
    idx = torch.randint(0, n_item, (B,), device=dev)
    pairs = B * n_item
    out = {'bound': 'mfma', 'unit': 'TFLOP/s',
           'workload': 'cfg-3 item term: cal_infonce_loss(items1[poss], items2[poss], items2, %.1f), B=%d anchors x M=%d rows, d=%d' % (temp, B, n_item, d),
           'timing': 'HIP events around the call(s), median of 10; preparation / finishing launches of a call included', 'modes': {}}
    for prec, peak, unit in (('h3', MFMA_BF16_PEAK_TF, 'fp16'), ('x6', MFMA_BF16_PEAK_TF, 'bf16'), ('fp32', MFMA_F32_PEAK_TF, 'fp32')):
        def fwd_nograd():
            with torch.no_grad():
                ops.infonce_loss_gathered(t1, t2, idx, temp, precision=prec)

        def fwd():
            ops.infonce_loss_gathered(t1, t2, idx, temp, precision=prec)

        def fb():
            t1.grad = t2.grad = None
            ops.infonce_loss_gathered(t1, t2, idx, temp, precision=prec).backward()
        ms_n, ms_f, ms_fb = time_events(fwd_nograd, 10), time_events(fwd, 10), time_events(fb, 10)
        code = ops.INFONCE_PRECISIONS[prec] << 8
        issued = sum(ops.infonce_issued_flops(k, B, n_item, d, code | ops.INFONCE_FWD_W_BIT)[0] for k in ('fwd', 'bwd'))
        issued_f = ops.infonce_issued_flops('fwd', B, n_item, d, code)[0]
        out['modes'][prec] = {
            'datatype': unit, 'peak': peak,
            'fwd_no_grad_ms': ms_n, 'fwd_no_grad_frac': issued_f / (ms_n * 1e-3) / 1e12 / peak, 'fwd_pairs_per_s': pairs / (ms_n * 1e-3),
            'fwd_differentiated_ms': ms_f, 'fwdbwd_ms': ms_fb, 'fwdbwd_pairs_per_s': pairs / (ms_fb * 1e-3),
            'issued_flops_fwdbwd': issued, 'achieved': issued / (ms_fb * 1e-3) / 1e12, 'frac': issued / (ms_fb * 1e-3) / 1e12 / peak,
            'fp32_equivalent_flops_fwdbwd': 8.0 * pairs * d, 'fp32_equivalent_TFLOPs': 8.0 * pairs * d / (ms_fb * 1e-3) / 1e12}
    head = out['modes']['h3']
    out.update({'mode': 'h3 (library default since round 5: two fp16 planes, three fp16-MFMA terms per product -- HALF the matrix instructions of x6, '
                        'so its fraction of the 2.5 PFLOP/s fp16 peak on ISSUED flops is lower while the call is 1.5x faster; fp32-equivalent rate beside it)',
                'achieved': head['achieved'], 'peak': head['peak'], 'frac': head['frac'], 'traffic': None,
                'fp32_equivalent_TFLOPs': head['fp32_equivalent_TFLOPs'], 'fp32_equivalent_over_fp32_mfma_peak': head['fp32_equivalent_TFLOPs'] / MFMA_F32_PEAK_TF,
                'kernel': 'infonce_bwd_lds_kernel<..., F16>: anchor-gradient role + row sums (forward call), all-gradient role (backward call)'})
    return out


def predict_multi_gpu(rows, cols, vals, n, graph, d, L, B, dev, step_ms, launch_us, edges_per_step):
    """What an N = 2 / 4 / 8 run of this command should print, so that a SCALE record is held against something.  One GPU can measure
    one rank's SHARE of either decomposition; what crosses GPUs is modelled (no link exists on a one-GPU box) and says so:
      feature-sliced (the N > 1 headline): the six fused launches of the propagation at d / N columns are measured here (same graph,
        same kernels: spmm_swept_kernel<32|16|8>); the step's other kernels (BPR, scatter, regularizer: this line's step minus its
        six launches) do not shrink with N; the one [3B, d/N] all-gather is priced at 25 us (latency-bound, < 1 MB) and the two hipGraph
        replays of the captured step at 12 us each (MI355X_MICROARCH.md: graph-replay-floor 10-16 us);
      row-sharded (BASELINE.json's wording): rank 0's shard matrix of N is built and its local product over the gathered table measured
        here; every all-gather hands a rank (N-1)/N of the 36.9 MB table, one shard per xGMI link in parallel, priced at 50-75 GB/s per
        link + 20 us; 2 L - 1 of them per step, not overlapped (what shard.py's all_gather mode issues)."""
    from sslrec_amd import ops
    from sslrec_amd.shard import ShardedGraph, rows_per_rank
    out = {'measured_on': 'this GPU, one rank\'s share; collectives modelled (see `model`)', 'n1_ms_per_step': step_ms, 'predictions': {}}
    other_ms = max(step_ms - 2 * L * launch_us * 1e-3, 0.0)
    out['model'] = {'non_spmm_ms_per_step_does_not_shrink': other_ms, 'feature_all_gather_us': 25.0, 'graph_replays_us': 24.0,
                    'row_all_gather_link_GBps': [50.0, 75.0], 'row_all_gather_latency_us': 20.0, 'row_all_gathers_per_step': 2 * L - 1}
    for N in (2, 4, 8):
        w = d // N
        pred = {}
        if d % N == 0 and w in (8, 16, 32):
            e0 = torch.randn(n, w, device=dev, requires_grad=True)
            gt = torch.randn(n, w, device=dev)

            def fb():
                e0.grad = None
                ops.propagate_sum(graph, e0, L).backward(gt)
            prop_ms = time_events(fb, 10, warmup=3)
            ms = prop_ms + other_ms + 0.025 + 0.024
            pred['feature_sliced'] = {'columns_per_gpu': w, 'propagation_fwd_bwd_ms_measured': prop_ms, 'ms_per_step': ms,
                                      'value_edges_per_s': edges_per_step / (ms * 1e-3), 'speedup_vs_n1': step_ms / ms}
            del e0, gt
        try:
            sg = ShardedGraph(rows, cols, vals, n, N, 0, dev)
            n_per = rows_per_rank(n, N)
            xg = torch.randn(n_per * N, d, device=dev)
            acc = torch.randn(n_per, d, device=dev)
            out_ = torch.empty_like(acc)
            prod_ms = time_events(lambda: ops.spmm_raw(sg.a, xg, 'fwd', acc_in=acc, acc_out=out_, want_y=True), 10, warmup=3)
            shard_mb = n_per * d * 4 / 1e6
            ag_ms = [shard_mb / 1e3 / gb * 1e3 + 0.020 for gb in (75.0, 50.0)]
            ms_lo, ms_hi = [2 * L * prod_ms + other_ms + (2 * L - 1) * a for a in ag_ms]
            pred['row_sharded'] = {'local_product_ms_measured': prod_ms, 'all_gather_ms_modelled': ag_ms, 'shard_MB_per_link': shard_mb,
                                   'ms_per_step': [ms_lo, ms_hi], 'value_edges_per_s': [edges_per_step / (ms_hi * 1e-3), edges_per_step / (ms_lo * 1e-3)],
                                   'speedup_vs_n1': [step_ms / ms_hi, step_ms / ms_lo]}
            # The third decomposition one might try: tables REPLICATED, only the rows of A split, one all-gather of the
            # product's rows per layer, overlapped as well as the dependencies allow.  Layer l + 1 gathers what layer l wrote, so a gather can
            # only hide under the product that CONSUMES it (own block first, peers' blocks as they land: the pipelined form) -- per layer
            # max(product, all-gather), never their difference -- and the replicated parameters need one more gather (the gradient) and an
            # optimizer pass over the whole table on every GPU.  Best case with the figures above:
            best = [(2 * L) * max(prod_ms, a) + other_ms for a in ag_ms]
            pred['rows_of_A_split_replicated_tables'] = {
                'ms_per_step_best_case': best, 'speedup_vs_n1_best_case': [step_ms / b for b in best],
                'why': 'same bytes on the wire as row_sharded (a gather of the layer table per product); with perfect overlap each of the 2 L '
                       'gathers still costs max(local product %.3f ms, all-gather %.3f-%.3f ms); the feature-sliced step has NO per-layer collective, '
                       'which is why it is the N > 1 headline at this table size' % (prod_ms, ag_ms[0], ag_ms[1])}
            del sg, xg, acc, out_
        except Exception as exc:
            pred['row_sharded'] = {'error': repr(exc)[:200]}
        torch.cuda.empty_cache()
        out['predictions']['n_gpus=%d' % N] = pred
    return out


def rccl_check_child(args):
    """The N > 1 code path of this file on a ONE-rank RCCL process group (`SSLREC_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: both
    decompositions, every collective really issued on backend "nccl" = RCCL), run as a child of the default N = 1 run so that the RCCL
    path executes in the driver's own bench run every round.  Returns the child's transport proof + step times; an
    execution check, not a scaling number."""
    import subprocess
    if os.environ.get('SSLREC_BENCH_CHILD') == '1':
        return {'skipped': 'child run'}
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-extras',
           '--no-configs', '--no-live-traffic', '--workload', args.workload, '--dim', str(args.dim), '--layers', str(args.layers)]
    env = dict(os.environ, SSLREC_BENCH_CHILD='1', SSLREC_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    try:
        t0 = time.perf_counter()
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, check=True)
        child = json.loads(res.stdout.decode().strip().splitlines()[-1])
        mg = child.get('multi_gpu', {})
        proof = mg.get('transport_proof', {})
        return {'command': 'SSLREC_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 6 --warmup 2 (child of this run)', 'wall_s': time.perf_counter() - t0,
                'backend': proof.get('backend'), 'is_rccl': proof.get('is_rccl'),
                'ranks_that_completed_an_all_reduce': proof.get('ranks_that_completed_an_all_reduce'),
                'devices_at_launch': proof.get('devices_at_launch'), 'collectives': proof.get('collectives'),
                'ms_per_step_by_decomposition': {k: v.get('ms_per_step') for k, v in child.get('decompositions', {}).items()},
                'rccl_ranks': {k: v.get('rccl_ranks') for k, v in child.get('decompositions', {}).items()},
                'note': 'process group of ONE rank with the one-rank short cuts off (SSLREC_FORCE_COLLECTIVES): every collective of the N > 1 '
                        'path executed on RCCL; the rates are those of a collective with itself'}
    except Exception as exc:
        return {'error': repr(exc)[:400]}


def measure_traffic_live(args):
    """`roofline.traffic` measured IN this run when rocprofv3 is on the box: two more runs of this command under
    `rocprofv3 --pmc <counter> --kernel-trace` (one counter per pass, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not
    fit one pass), a few steps each, dense launches only (SSLREC_SPARSE_GRAD=0, like the timed region); bytes per launch of the
    dominant SpMM kernel = FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes) + WRITE_SIZE (as reported: uncalibrated).
    Returns (bytes, description) or (None, reason); the caller falls back to the stamped file of the last profiled commit."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get('SSLREC_BENCH_CHILD') == '1' or args.no_live_traffic:
        return None, 'switched off'
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None:
        return None, 'rocprofv3 not found'
    means = {}
    tmp = tempfile.mkdtemp(prefix='sslrec_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out_dir = os.path.join(tmp, counter)
            cmd = [prof, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out_dir, '-o', 'p', '--', sys.executable,
                   os.path.abspath(__file__), '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-extras', '--no-configs', '--no-live-traffic',
                   '--workload', args.workload, '--dim', str(args.dim), '--layers', str(args.layers)]
            env = dict(os.environ, SSLREC_BENCH_CHILD='1', SSLREC_SPARSE_GRAD='0', TMPDIR='/tmp')
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180, check=True)
            vals = []
            for f in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if 'spmm_swept_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                        vals.append(float(r['Counter_Value']))
            if not vals:
                return None, 'no %s samples of the SpMM kernel in the rocprofv3 output' % counter
            means[counter] = (float(np.mean(vals)), len(vals))
    except Exception as exc:
        return None, 'rocprofv3 pass failed: %r' % (exc,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = means['FETCH_SIZE'][0] * 1024 * 2, means['WRITE_SIZE'][0] * 1024
    return fetch + write, ('measured in THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (6 steps each, %d / %d launches of the '
                           'kernel, dense launches): FETCH_SIZE x 2 (gfx950) = %.1f MB + WRITE_SIZE = %.1f MB per launch'
                           % (means['FETCH_SIZE'][1], means['WRITE_SIZE'][1], fetch / 1e6, write / 1e6))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher (how the driver may call it): start the N ranks here, one process per
    GPU, rendezvous on 127.0.0.1; rank 0 inherits stdout and prints the one JSON line; a failing rank fails the run."""
    import socket
    import subprocess
    n = args.gpus
    one_device = os.environ.get('SSLREC_BENCH_ONE_DEVICE') == '1'
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (no CPU fallback)')
    if not one_device and torch.cuda.device_count() < n:
        sys.exit('bench.py --gpus %d: only %d GPU(s) visible (SSLREC_BENCH_ONE_DEVICE=1 runs the N > 1 code path on one '
                 'device over gloo, for checking the code -- not the speed)' % (n, torch.cuda.device_count()))
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs and rc == 0:
            time.sleep(0.2)
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = code
    finally:
        for p in procs:           # a rank failed (or we were interrupted): the others would wait in a collective for ever
            p.kill()
        for p in procs:
            p.wait()
    sys.exit(rc)


def timed_steps(step, steps, warmup, barrier, before_timed=None):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize on both sides; wall seconds of the K steps"""
    for _ in range(warmup):
        step()
    barrier()
    if before_timed is not None:
        before_timed()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def launches_summary(records):
    """records: [(plan, d, has_acc, want_y, seconds)] of SpMM launches -> (mean seconds, mean algorithmic bytes, kernel name)"""
    secs = [r[4] for r in records]
    def extra(r):       # swept-layout launches: the zero-row hint's row count, the deferred layer tables the launch adds up
        kw = {}
        if len(r) > 5 and r[5] is not None:
            kw['x_rows'] = r[5]
        if len(r) > 6 and r[6]:
            kw['sum_in'] = r[6]
        return kw
    byts = [r[0].algorithmic_bytes(r[1], acc=r[2], write_y=r[3], **extra(r)) for r in records]
    launches_summary.own_bytes = float(np.mean([r[0].algorithmic_bytes(r[1], acc=r[2], write_y=r[3], pattern=True, **extra(r))
                                                if (len(r) > 7 and r[7]) else b for r, b in zip(records, byts)]))
    launches_summary.pattern_share = float(np.mean([1.0 if (len(r) > 7 and r[7]) else 0.0 for r in records]))
    plan, dd = records[0][0], records[0][1]
    name = ('spmm_swept_kernel<%d> (LDS accumulators, column-swept)' % getattr(plan, 'width', dd)) if type(plan).__name__ == 'SweptLayout' \
        else 'spmm_stream_kernel<%d> (+long-row reduce)' % dd
    return float(np.mean(secs)), float(np.mean(byts)), name


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
    ap.add_argument('--no-live-traffic', action='store_true', help='do not run the two rocprofv3 --pmc passes behind roofline.traffic (the stamped file is used)')
    # N > 1: 'feature' (default headline) = every GPU holds all rows and d / N columns, no collective in the propagation
    # (sslrec_amd/feature_shard.py); the others = row-sharded tables with one exchange per layer (sslrec_amd/shard.py).
    # Whatever the headline, the OTHER family is timed in the same run and reported under multi_gpu (feature <-> all_gather).
    ap.add_argument('--shard-mode', default='feature', choices=['feature', 'all_gather', 'pipelined', 'reduce_scatter'])
    ap.add_argument('--no-second-decomposition', action='store_true', help='N > 1: time only the headline decomposition')
    ap.add_argument('--config', default='cfg2', choices=['cfg1', 'cfg2', 'cfg3', 'cfg4'],
                    help='BASELINE.json config: cfg2 (default) = the headline LightGCN / amazon-book line; cfg1 / cfg3 / cfg4 = the other '
                         'single-GPU configs through the model classes (bench_configs.py); the default run carries all three under `configs`')
    ap.add_argument('--no-configs', action='store_true', help='default run: skip the cfg1 / cfg3 / cfg4 lines')
    ap.add_argument('--eager-step', action='store_true',
                    help='feature mode: issue the step as ~40 eager launches instead of two captured hipGraphs around the all-gather')
    args = ap.parse_args()

    if args.config != 'cfg2':
        if args.gpus != 1 or not torch.cuda.is_available():
            sys.exit('bench.py --config %s is a single-GPU line and needs a GPU' % args.config)
        from bench_configs import run_config
        print(json.dumps(run_config(args.config, args.steps, args.warmup, with_cpu=not args.no_cpu_baseline)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args)                 # does not return
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints "Librccl path : ..." through C stdio, which
    # a pipe delivers at exit, i.e. AFTER the line): keep the real stdout for the line and point descriptor 1 at stderr for everybody else.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        sys.exit('bench.py --gpus %d inside a job of WORLD_SIZE=%d: launch it with --nproc-per-node %d, or without a launcher '
                 '(it starts its own ranks)' % (args.gpus, world, args.gpus))
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
    # SSLREC_BENCH_FORCE_DIST=1 (with --gpus 1): the N > 1 code path -- both decompositions, every collective really issued
    # (SSLREC_FORCE_COLLECTIVES) -- on a process group of ONE rank: how the RCCL calls are exercised on a box with one GPU.
    force_dist = world == 1 and os.environ.get('SSLREC_BENCH_FORCE_DIST') == '1'
    dist_path = world > 1 or force_dist
    if force_dist:
        os.environ['SSLREC_FORCE_COLLECTIVES'] = '1'
        if 'MASTER_PORT' not in os.environ:
            import socket
            sock = socket.socket()
            sock.bind(('127.0.0.1', 0))
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(sock.getsockname()[1]), RANK='0', WORLD_SIZE='1')
            sock.close()
    if dist_path:
        if one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(dev))

    from sslrec_amd import ops
    d, L = args.dim, args.layers
    trn, rows, cols, vals, n = build_graph_host(args.workload)
    ue, ie = xavier_tables(trn.shape[0], trn.shape[1], d)
    e0_full = torch.cat([ue, ie])

    B = 4096                                                   # train.batch_size of every target yml
    bgen = torch.Generator().manual_seed(11)
    batch = [torch.randint(0, trn.shape[0], (B,), generator=bgen), torch.randint(0, trn.shape[1], (B,), generator=bgen),
             torch.randint(0, trn.shape[1], (B,), generator=bgen)]
    batch = [b.to(dev) for b in batch]
    reg_weight = 1.0e-8
    edges_per_step = 2 * L * int(vals.size)

    def barrier():
        if dist_path:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_path:
            return x
        t_ = torch.tensor([x], dtype=torch.float64, device='cpu' if one_device else dev)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    def run_eager(step):
        """K eager steps with HIP event pairs around the SpMM launches of the timed region.  An event record is a packet of its own on
        the stream -- a pair around EVERY launch costs the step 35-40 us of bubbles (tools/eager_overhead.py: 0.519 against 0.479 ms
        without events) -- so a pair goes around every n-th launch, n the first of 5, 7, 9, ... coprime with the launches of a step:
        the sampled launch rotates through all of them (L = 3: six launches, every 5th = 24 pairs in 20 steps, four per launch)."""
        def arm():
            ops.PROFILE, ops.PROFILE_EVERY = [], every
        # (launches of a step: 2 L products on one GPU; the sharded steps add their own -- the list of the timed region says how many)
        every = 5
        while math.gcd(every, 2 * L) != 1:
            every += 2
        if args.steps * 2 * L < 4 * every:      # (a run of a few steps: every launch)
            every = 1
        elapsed = timed_steps(step, args.steps, args.warmup, barrier, before_timed=arm)
        prof, ops.PROFILE, ops.PROFILE_EVERY = ops.PROFILE, None, 1
        run_eager.n_launches = len(prof)
        recs = [(r[2], r[3], r[4], r[5], r[0].elapsed_time(r[1]) * 1e-3, r[7] if len(r) > 7 else None, r[8] if len(r) > 8 else 0,
                 bool(r[9]) if len(r) > 9 else False)
                for r in prof if r[0] is not None]
        # the same timings by the launch's position inside its step (forward layers, then backward layers): where a launch's operand
        # comes from matters (EXPERIMENTS.md C.2)
        per_step = len(prof) // max(1, args.steps)
        run_eager.by_position = None
        if per_step > 0 and len(prof) == per_step * args.steps:
            pos = {}
            for i, r in enumerate(prof):
                if r[0] is not None:
                    pos.setdefault(i % per_step, []).append(r[0].elapsed_time(r[1]) * 1e3)
            run_eager.by_position = [round(float(np.mean(pos[k])), 2) if k in pos else None for k in range(per_step)]
        return max_over_ranks(elapsed), recs, ('HIP events around every SpMM launch of the timed region' if every == 1 else
                                               'HIP events around every %dth SpMM launch of the timed region (%d of its %d launches; the sampled launch '
                                               'rotates through the %d of a step)' % (every, len(recs), len(prof), len(prof) // max(1, args.steps)))

    graph = None
    results = {}               # decomposition -> dict(elapsed, recs, timing, step kind)
    if not dist_path:
        from sslrec_amd.graph import PropGraph
        graph = PropGraph(rows, cols, vals, (n, n), dev)
        e0 = e0_full.to(dev).requires_grad_(True)
        one = torch.ones((), dtype=torch.float32, device=dev)      # d loss / d loss, made once (backward() would fill a new one per step)

        def step():     # LightGCN cal_loss + backward (lightgcn.py:45-56) on the fused path, keep_rate 1.0
            e0.grad = None
            s, reg = ops.propagate_sum(graph, e0, L, reg_weight=reg_weight)      # (the regularizer's gradient rides on the last backward product)
            loss, _bpr = ops.bpr_loss_stacked(s, trn.shape[0], *batch, divisor=B, add=reg)      # bpr + reg (lightgcn.py:54) from the BPR kernel's finishing step
            loss.backward(one)
        # The headline counts 2 L nnz propagated edges per step, so every one of them is really multiplied: the library's default of
        # telling the first backward product which rows of the BPR gradient are zero (ops.SPARSE_GRAD, same result, that launch 80 -> 55 us)
        # is switched OFF for the timed region and reported separately below.
        sparse_default, ops.SPARSE_GRAD = ops.SPARSE_GRAD, False
        elapsed, recs, timing = run_eager(step)
        results['single'] = dict(elapsed=elapsed, recs=recs, timing=timing, graphed=False, n_launches=run_eager.n_launches,
                                 by_position=run_eager.by_position)
        headline = 'single'
        ops.SPARSE_GRAD = sparse_default
        hint_elapsed, hint_recs = 0.0, []
        if sparse_default:      # (SSLREC_SPARSE_GRAD=0, e.g. the profiling passes of tools/gpu_profile.sh: dense launches only)
            hint_elapsed, hint_recs, _ = run_eager(step)
        hinted = [r for r in hint_recs if r[5] is not None]
        # the same step replayed as ONE captured hipGraph (no Python, no event records between the launches), without and with the hint
        graphed = {}
        try:
            for label, flag in ((('dense', False), ('zero_row_hint', True)) if sparse_default else (('dense', False),)):
                ops.SPARSE_GRAD = flag
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                cg = torch.cuda.CUDAGraph()
                e0.grad = None
                with torch.cuda.graph(cg):
                    step()
                graphed['ms_per_step_' + label] = timed_steps(cg.replay, args.steps, args.warmup, barrier) / args.steps * 1e3
                del cg
        except Exception as exc:
            graphed['error'] = repr(exc)[:300]
        ops.SPARSE_GRAD = sparse_default
        results['single']['as_one_hip_graph'] = graphed
        results['single']['zero_row_hint'] = None if not hinted else {
            'ms_per_step': hint_elapsed / args.steps * 1e3,
            'hinted_launch_us': float(np.mean([r[4] for r in hinted])) * 1e6 if hinted else None,
            'hinted_launch_algorithmic_bytes': launches_summary(hinted)[1] if hinted else None,
            'note': 'the same K steps with the library default ops.SPARSE_GRAD on: the first backward product of a step skips the entries '
                    'of the rows the BPR loss did not write (<= 3B of N); bit-identical gradients'}
    else:
        feature_ok = d % world == 0 and d // world in (8, 16, 32, 64, 128, 256)
        headline = args.shard_mode
        if headline == 'feature' and not feature_ok:
            headline = 'all_gather'          # d / N is not a width of the kernel: row shards
        modes = [headline]
        if not args.no_second_decomposition:
            other = 'all_gather' if headline == 'feature' else 'feature'
            if other != 'feature' or feature_ok:
                modes.append(other)
        for mode in modes:
            if mode == 'feature':
                from sslrec_amd.feature_shard import FeatureSlicedGraphCF, GraphedLightGCNStep
                from sslrec_amd.graph import PropGraph
                graph = PropGraph(rows, cols, vals, (n, n), dev)
                model = FeatureSlicedGraphCF(graph, trn.shape[0], trn.shape[1], e0_full, L, world, rank)

                def eager_step(model=model):     # local propagation of d / N columns, batch rows by one all-gather
                    model.local_embeds.grad = None
                    model.lightgcn_loss(batch, reg_weight).backward()
                graphed = False
                if not args.eager_step:      # two hipGraph replays around the one collective: a GPU's share of the work is smaller
                    stamps = ops.StampLog(dev)                                  # than the host time of the eager launches
                    try:
                        gstep = GraphedLightGCNStep(model, B, reg_weight, stamps=stamps)
                        captured = 1
                    except Exception as exc:                          # capture refused on this box: every rank falls back to eager launches
                        sys.stderr.write('rank %d: hipGraph capture failed (%r); eager step\n' % (rank, exc))
                        captured = 0
                    flag = torch.tensor([captured], dtype=torch.int32, device='cpu' if one_device else dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    graphed = bool(flag.item())
                if graphed:
                    elapsed = timed_steps(lambda: gstep.step(batch), args.steps, args.warmup, barrier, before_timed=stamps.reset_counts)
                    recs = [(m[0], m[1], m[2], m[3], ms * 1e-3, None, m[6] if len(m) > 6 else 0) for m, ms, cnt in stamps.read() for _ in range(1)]
                    n_exec = [cnt for _, _, cnt in stamps.read()]
                    assert all(c == args.steps for c in n_exec), n_exec       # every captured launch ran once per timed step
                    results[mode] = dict(elapsed=max_over_ranks(elapsed), recs=recs * args.steps, graphed=True,
                                         timing='device wall clock inside every SpMM launch of the timed region (first workgroup start to last '
                                                'workgroup end, accumulated over the replays; HIP events cannot be recorded inside a captured hipGraph)')
                else:
                    elapsed, recs, timing = run_eager(eager_step)
                    results[mode] = dict(elapsed=elapsed, recs=recs, timing=timing, graphed=False, n_launches=run_eager.n_launches)
                del model
            else:
                from sslrec_amd.shard import ShardedGraph, ShardedGraphCF
                sg = ShardedGraph(rows, cols, vals, n, world, rank, dev)
                model = ShardedGraphCF(sg, trn.shape[0], trn.shape[1], e0_full, L, mode=mode)

                def step(model=model):     # same step on row-sharded tables: collectives through autograd
                    model.local_embeds.grad = None
                    model.lightgcn_loss(batch, reg_weight).backward()
                elapsed, recs, timing = run_eager(step)
                results[mode] = dict(elapsed=elapsed, recs=recs, timing=timing, graphed=False, n_launches=run_eager.n_launches)
                del model, sg
            torch.cuda.empty_cache()

    head = results[headline]
    elapsed = head['elapsed']
    value = edges_per_step * args.steps / elapsed

    # roofline of the dominant kernel from the per-launch timings of the timed region (this rank)
    avg_s, avg_bytes, kname = launches_summary(head['recs'])
    own_bytes, pattern_share = launches_summary.own_bytes, launches_summary.pattern_share
    achieved = avg_bytes / avg_s / 1e9
    traffic = traffic_src = None
    tf = os.path.join(ROOT, 'profiles', 'spmm_traffic.json')
    live_note = None
    if not dist_path and rank == 0 and not args.no_extras:      # (the quick experiment lines of the tools skip it with --no-extras)
        traffic, traffic_src = measure_traffic_live(args)
        if traffic is None:
            live_note, traffic_src = traffic_src, None
    if traffic is None and os.path.exists(tf) and not dist_path:       # (the stamp belongs to the single-GPU kernel) PMC passes are separate runs (rocprofv3 --pmc): the committed file of the last profiled commit
        tj = json.load(open(tf))
        traffic = tj.get('hbm_bytes_per_launch')
        traffic_src = 'profiles/spmm_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of this command, round %s, commit %s%s' % (
            tj.get('measured_in_round'), tj.get('measured_at_commit'), '' if live_note is None else ' (not measured in this run: %s)' % live_note)
    roofline = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src, 'kernel': kname,
                'avg_launch_us': avg_s * 1e6, 'launches': head.get('n_launches', len(head['recs'])), 'launches_timed': len(head['recs']),
                'launch_timing': head['timing'],
                'algorithmic_bytes_per_launch': avg_bytes}
    if pattern_share > 0:      # factorized normalization (ops.FACTORIZED): most launches of a layer chain read no value array
        roofline['factorized_chain'] = {
            'pattern_launch_share': pattern_share, 'bytes_these_launches_have_to_move': own_bytes,
            'frac_on_those_bytes': own_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
            'note': '`achieved` / `frac` price every launch at SURVEY 8d\'s bytes of the OPERATION (entries * 8: column + value), as in rounds 1-4; a '
                    'pattern launch of the factorized chain (values = r[i] * r[j]: the scaled table is gathered, the row sum scaled in the flush) '
                    'reads entries * 4 + one factor per row -- the stricter figure is given here'}
    if traffic:      # what the memory system actually moved per launch (L2 <-> fabric requests by the PMC counters) over the same launch time
        roofline['traffic_GBps'] = traffic / avg_s / 1e9
        roofline['traffic_frac_of_peak'] = traffic / avg_s / 1e9 / HBM_PEAK_GBS
        roofline['traffic_over_algorithmic'] = traffic / avg_bytes
    roofline['achieved_is'] = ('operation-equivalent: SURVEY 8d bytes of the operation (entries * 8 + pointers + X once + Y once [+ 2 Y with the fused '
                               'accumulator]) over the measured launch time; the bytes a factorized chain has to move are in factorized_chain')
    if pattern_share > 0:
        roofline['achieved_on_bytes_moved'] = own_bytes / avg_s / 1e9
        roofline['frac_on_bytes_moved'] = own_bytes / avg_s / 1e9 / HBM_PEAK_GBS
    if not dist_path and rank == 0 and not args.no_extras:
        try:      # the ceiling of any one-gather-per-entry kernel on this graph, measured here and now
            roofline['ceiling'] = gather_ceiling(dev, trn.shape[0], trn.shape[1], d, int(vals.size), avg_bytes, avg_s)
        except Exception as exc:
            roofline['ceiling'] = {'error': repr(exc)[:300]}
    if head.get('by_position'):
        roofline['launch_us_by_position_in_step'] = head['by_position']
    if head.get('zero_row_hint'):
        roofline['with_zero_row_hint'] = head['zero_row_hint']
    if head.get('as_one_hip_graph'):
        roofline['step_as_one_hip_graph'] = head['as_one_hip_graph']
    hinted = [r for r in head['recs'] if len(r) > 5 and r[5] is not None]
    if hinted:      # the first backward product of every step is told which rows of the BPR gradient are not zero (ops.SPARSE_GRAD)
        dense = [r for r in head['recs'] if not (len(r) > 5 and r[5] is not None)]
        roofline['launch_kinds'] = {
            'dense': {'launches': len(dense), 'avg_launch_us': float(np.mean([r[4] for r in dense])) * 1e6,
                      'algorithmic_bytes': launches_summary(dense)[1]},
            'zero_row_hint': {'launches': len(hinted), 'avg_launch_us': float(np.mean([r[4] for r in hinted])) * 1e6,
                              'algorithmic_bytes': launches_summary(hinted)[1], 'nonzero_rows_at_most': int(hinted[0][5])}}

    # multi-GPU: per decomposition the local SpMM time, the step's collectives timed alone, the overlap (SURVEY.md §8e asks
    # for edges/s with and without the per-layer collective)
    multi = None
    if dist_path:
        from sslrec_amd.shard import all_gather_rows, all_reduce_sum, reduce_scatter_rows, rows_per_rank, shards_pipelined
        n_per = rows_per_rank(n, world)
        xs = torch.randn(n_per, d, device=dev)
        small = torch.zeros(3 * B, d, device=dev)

        def exchanges(mode):
            if mode == 'feature':          # the step's only collective: the [3B, d/N] slices of the batch rows
                all_gather_rows(small[:, :d // world].contiguous(), world)
                return
            for _ in range(2 * L - 1):     # L forward + L-1 backward exchanges of [n_per, d] rows, one [3B, d] all-reduce
                if mode == 'pipelined':
                    for _q, _x in shards_pipelined(xs, world, rank):
                        pass
                elif mode == 'reduce_scatter':
                    reduce_scatter_rows(torch.empty(n_per * world, d, device=dev), world)
                else:
                    all_gather_rows(xs, world)
            all_reduce_sum(small)

        def describe(mode):
            r = results[mode]
            for _ in range(3):
                exchanges(mode)
            barrier()
            t1 = time.perf_counter()
            for _ in range(10):
                exchanges(mode)
            barrier()
            coll_s = (time.perf_counter() - t1) / 10
            step_s = r['elapsed'] / args.steps
            local_s = float(np.mean([x[4] for x in r['recs']])) * r.get('n_launches', len(r['recs'])) / args.steps      # (sampled launches x all launches)
            k_s, k_b, k_name = launches_summary(r['recs'])
            return {'decomposition': mode, 'value_edges_per_s': edges_per_step / step_s, 'ms_per_step': step_s * 1e3,
                    'step': 'two hipGraph replays around the all-gather' if r['graphed'] else 'eager launches',
                    'local_spmm_ms': local_s * 1e3, 'collective_ms': coll_s * 1e3,
                    # share of the collective time that the step hides behind compute (0 when they simply add up; the
                    # step's other kernels -- BPR, regularizer -- make this a lower bound)
                    'overlap_frac': float(min(1.0, max(0.0, (local_s + coll_s - step_s) / coll_s))) if coll_s > 0 else None,
                    'edges_per_s_excluding_collective': edges_per_step / local_s,
                    'spmm_kernel': k_name, 'spmm_avg_launch_us': k_s * 1e6, 'spmm_hbm_roofline_frac': k_b / k_s / 1e9 / HBM_PEAK_GBS,
                    'launch_timing': r['timing'],
                    'collective_bytes_per_rank_per_layer': 0 if mode == 'feature' else int(n * d * 4 * (world - 1) / world),
                    'collective_bytes_per_rank_per_step': int(3 * B * d * 4 * (world - 1) / world) if mode == 'feature'
                    else int((2 * L - 1) * n * d * 4 * (world - 1) / world + 3 * B * d * 4)}
        multi = describe(headline)
        multi['collective'] = headline
        # --- what carried the bytes: the backend torch.distributed reports, the ranks that COMPLETED a collective on it
        # (an all-reduce of ones over the group on device memory: under RCCL the sum is the number of ranks whose kernel ran), the devices
        # the ranks sit on, and the rate of each collective of the step, alone, per rank and per xGMI link
        backend = dist.get_backend()
        proof = {'backend': backend, 'is_rccl': backend == 'nccl'}
        ones = torch.ones(1, device='cpu' if one_device else dev)
        dist.all_reduce(ones)
        proof['ranks_that_completed_an_all_reduce'] = int(ones.item())
        ids = [None] * world
        props = torch.cuda.get_device_properties(local_rank)
        dist.all_gather_object(ids, {'rank': rank, 'device': dev, 'name': props.name, 'pci_bus_id': getattr(props, 'pci_bus_id', None),
                                     'uuid': str(getattr(props, 'uuid', '')), 'pid': os.getpid()})
        proof['devices_at_launch'] = ids
        proof['distinct_devices'] = len({(i_['device'], i_['uuid']) for i_ in ids})

        def rate(fn, bytes_out_per_rank):
            for _ in range(2):
                fn()
            barrier()
            t1_ = time.perf_counter()
            for _ in range(5):
                fn()
            barrier()
            s_ = max_over_ranks((time.perf_counter() - t1_) / 5)
            links = max(1, world - 1)
            return {'ms': s_ * 1e3, 'bytes_sent_per_rank': int(bytes_out_per_rank), 'GBps_per_rank': bytes_out_per_rank / s_ / 1e9,
                    'GBps_per_link': bytes_out_per_rank / links / s_ / 1e9}
        shard_b = n_per * d * 4
        proof['collectives'] = {
            'all_gather_rows [n/N, d]': rate(lambda: all_gather_rows(xs, world), shard_b * (world - 1)),
            'reduce_scatter_rows [n, d]': rate(lambda: reduce_scatter_rows(torch.empty(n_per * world, d, device=dev), world), shard_b * (world - 1)),
            'all_reduce [3B, d]': rate(lambda: all_reduce_sum(small), 2 * small.numel() * 4 * (world - 1) / max(world, 1)),
            'all_gather batch slices [3B, d/N]': rate(lambda: all_gather_rows(small[:, :max(1, d // world)].contiguous(), world),
                                                      small.shape[0] * max(1, d // world) * 4 * (world - 1))}
        proof['note'] = ('GBps_per_link = bytes a rank sends / (N - 1) peers / time: xGMI is point to point, one link per peer; under gloo (one '
                         'device, host-staged) these rates describe the host, not a link')
        multi['transport_proof'] = proof
        multi['transport'] = 'gloo, host-staged, all ranks on ONE device (code check only: these numbers mean nothing)' if one_device \
            else ('RCCL (torch.distributed backend nccl), process group of ONE rank with every collective of the N > 1 path issued '
                  '(SSLREC_BENCH_FORCE_DIST: an execution check of the RCCL calls, not a scaling number)' if force_dist
                  else 'RCCL (torch.distributed backend nccl) over xGMI')
        for mode in results:
            if mode != headline:
                multi['row_sharded' if mode != 'feature' else 'feature_sliced'] = describe(mode)
    if rank == 0:
        graphed = head['graphed']
        line = {
            'metric': 'propagation_edges_per_sec', 'value': value, 'unit': 'edges/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # (front-loaded: a reader that keeps the first ~100 characters still sees which configuration the line is quoted on)
            'config': {'workload': 'cfg2 LightGCN %s d=%d L=%d B=%d keep=1.0 f32: cal_loss+backward (propagation fwd+bwd, fused BPR, reg) on the '
                                   '%s-shaped synthetic graph (%dx%d, E=%d, nnz=%d)' % (args.workload, d, L, B, args.workload, trn.shape[0], trn.shape[1],
                                                                                        trn.nnz, vals.size),
                       'edges_per_step': edges_per_step,
                       'parallelism': 'single GPU' if not dist_path else
                       ('feature-sliced: all rows x %d of %d embedding columns per GPU, whole adjacency on each of %d GPUs, no collective in '
                        'the propagation, one [3B, d/N] all-gather per step%s' % (d // world, d, world, '; step = two hipGraph replays around the all-gather' if graphed else '')) if headline == 'feature' else
                       'rows dealt cyclically over %d GPUs, one %s per layer' % (world, headline)},
            'roofline': roofline,
        }
        if multi is not None:
            line['multi_gpu'] = multi
            # both decompositions as equal top-level blocks, so that a 1 -> 8 curve reads the same whichever is the headline:
            # value_row_sharded = the partition BASELINE.json words (rows dealt over the GPUs, one all-gather per layer and direction),
            # value_feature_sliced = all rows x d / N columns per GPU (no collective in the propagation)
            blocks = {('feature_sliced' if multi['decomposition'] == 'feature' else 'row_sharded'): multi}
            for key in ('row_sharded', 'feature_sliced'):
                if key in multi:
                    blocks[key] = multi[key]
            line['decompositions'] = {}
            for key, blk in blocks.items():
                line['value_' + key] = blk['value_edges_per_s']
                line['decompositions'][key] = {
                    'value_edges_per_s': blk['value_edges_per_s'], 'ms_per_step': blk['ms_per_step'],
                    # ranks that completed a collective ON RCCL in this run (0 under gloo: a one-device code check)
                    'rccl_ranks': multi['transport_proof']['ranks_that_completed_an_all_reduce'] if multi['transport_proof']['is_rccl'] else 0,
                    'transport': multi['transport'], 'collective_ms_per_step_alone': blk['collective_ms'],
                    'collective_bytes_per_rank_per_step': blk['collective_bytes_per_rank_per_step'],
                    'local_spmm_ms_per_step': blk['local_spmm_ms'], 'overlap_frac': blk['overlap_frac'],
                    'edges_per_s_excluding_collective': blk['edges_per_s_excluding_collective'], 'step': blk['step'],
                    'spmm_hbm_roofline_frac': blk['spmm_hbm_roofline_frac']}
            line['headline_decomposition'] = 'feature_sliced' if multi['decomposition'] == 'feature' else 'row_sharded (%s)' % multi['decomposition']
        if not dist_path and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(rows, cols, vals, n, d)
        if not dist_path and not args.no_extras:
            try:
                line['extras'] = extras_single_gpu(trn, rows, cols, vals, n, graph, d, L, dev)
            except Exception as exc:                      # extras never invalidate the headline
                line['extras'] = {'error': repr(exc)}
        if not dist_path and not args.no_extras and args.workload == 'amazon-book':
            try:      # what N = 2 / 4 / 8 should print: measured one-rank shares + modelled collectives
                line['multi_gpu_predicted'] = predict_multi_gpu(rows, cols, vals, n, graph, d, L, B, dev, elapsed / args.steps * 1e3,
                                                                avg_s * 1e6, edges_per_step)
            except Exception as exc:
                line['multi_gpu_predicted'] = {'error': repr(exc)[:300]}
        if not dist_path and not args.no_extras:
            line['rccl_check'] = rccl_check_child(args)
            try:
                l2 = (roofline.get('ceiling') or {}).get('gather_TBps_L2_resident') or 26.0
                line.setdefault('extras', {})['roofline_other_graphs'] = other_graphs_roofline(d, dev, l2)
            except Exception as exc:
                line.setdefault('extras', {})['roofline_other_graphs'] = {'error': repr(exc)[:300]}
        if not dist_path and not args.no_extras:
            try:      # the fused InfoNCE as a block of its own: half of BASELINE.json's metric (InfoNCE pairs/s) and the dominant kernel of cfg 3 / cfg 4
                line['roofline_infonce'] = infonce_roofline(trn.shape[1], d, dev)
            except Exception as exc:
                line['roofline_infonce'] = {'error': repr(exc)}
        if not dist_path and not args.no_configs and args.workload == 'amazon-book':
            del graph
            torch.cuda.empty_cache()
            from bench_configs import run_config
            line['configs'] = {}
            for tag in ('cfg1', 'cfg3', 'cfg4'):      # the other single-GPU configurations of BASELINE.json, ~30 steps each
                try:
                    line['configs'][tag] = run_config(tag, 30, 5, dev, with_cpu=not args.no_cpu_baseline)
                except Exception as exc:
                    line['configs'][tag] = {'error': repr(exc)[:300]}
        if not dist_path and not args.no_configs and not args.no_extras and args.workload == 'amazon-book':
            try:      # an EPOCH of the Trainer at this size: every default (bit parity with the reference's batches and draws) against every opt-in
                from bench_configs import epoch_times
                line.setdefault('extras', {})['epoch'] = epoch_times(dev)
                for k_ in ('epoch_s_default', 'epoch_s_fast', 'epoch_s_python'):
                    line['extras'][k_] = line['extras']['epoch'].get(k_)
            except Exception as exc:
                line.setdefault('extras', {})['epoch'] = {'error': repr(exc)[:300]}
        line_out.write(json.dumps(line) + '\n')
        line_out.flush()
    if dist_path:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
