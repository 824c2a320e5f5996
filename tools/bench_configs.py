#!/usr/bin/env python
"""One JSON line per BASELINE.json config that fits one GPU besides the headline (cfg 2 is bench.py's default line):
  cfg1  LightGCN step on the gowalla-shaped synthetic graph (the train pickle is missing upstream), d=32, L=2
  cfg3  SimGCL step on the amazon-book-shaped graph, d=64, L from simgcl.yml (2)
  cfg4  SGL-ED step on the REAL yelp interactions (tests/golden/yelp_lightgcn_d64_L2.npz), d=64, L from sgl.yml (2), keep 0.5
Each line: step time (cal_loss + backward through the model classes, perf-mode RNG), propagated (kept) directed edges per second,
and the HBM roofline of the SpMM launches inside the step (HIP events on the launch stream, algorithmic bytes of SURVEY.md §8d)
-- the conventions of bench.py, which runs one of them with `python bench.py --config cfgN`.
usage: python tools/bench_configs.py [--steps 30] > profiles/rNN/configs.jsonl"""
import argparse, json, os, sys
import numpy as np
import scipy.sparse as sp, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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


def run_config(tag, steps=30, warmup=5, dev='cuda:0'):
    from sslrec_amd import ops
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
    from sslrec_amd.data_utils.synth import make_dataset
    from sslrec_amd.models.bulid_model import build_model
    model_name, graph_name, d, L, extra, desc = CONFIGS[tag]
    raw = yelp_real() if graph_name == 'yelp-real' else make_dataset(graph_name)
    trn = sp.coo_matrix((raw != 0).astype(np.float32))          # what DataHandlerGeneralCF._load_one_mat does to a pickle
    over = {'data': {'synthetic': 'tiny'}, 'model': {'embedding_size': d, 'device_rng': True}}
    if L is not None:
        over['model']['layer_num'] = L
    over['model'].update(extra)
    load_config(model_name, device=dev, overrides=over)
    L = configs['model']['layer_num']
    dh = DataHandlerGeneralCF()
    dh.trn_mat = trn
    configs['data']['user_num'], configs['data']['item_num'] = trn.shape
    dh.torch_adj = dh._make_torch_adj(trn).to(dev)
    torch.manual_seed(0)
    model = build_model(dh).to(dev)
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
    ops.PROFILE = []
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    # the same step as ONE captured hipGraph (what `train.hip_graph` does for the Trainer): launch-bound steps -- cfg 1's 17 small launches
    # take 0.12 ms of GPU time and 0.4 ms of Python -- show what the device is left with
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
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / steps * 1e3
    except Exception as exc:      # (a step that cannot be captured keeps its eager line)
        graph_err = repr(exc)[:300]
    k_ms = [a.elapsed_time(b) for a, b, *_ in prof]
    # (a launch told which rows of its operand are zero -- LightGCN's first backward product -- reads only those: rec[7])
    k_bytes = [r[2].algorithmic_bytes(r[3], acc=r[4], write_y=r[5], **({'x_rows': r[7]} if len(r) > 7 and r[7] is not None else {}))
               - (1.0 - r[6]) * r[2].nnz * 8 for r in prof]
    edges = float(np.sum([r[2].nnz * r[6] for r in prof])) / steps
    ach = float(np.sum(k_bytes)) / (float(np.sum(k_ms)) * 1e-3) / 1e9
    return {'metric': 'propagation_edges_per_sec', 'value': edges * steps / elapsed, 'unit': 'edges/s', 'n_gpus': 1, 'steps': steps,
            'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'real yelp interactions' if graph_name == 'yelp-real' else 'synthetic',
            'config': {'workload': '%s: %s cal_loss+backward, %dx%d, %d interactions (%d directed entries), d=%d, L=%d, B=%d, augmentation '
                                   'randomness computed in the kernels (model.device_rng)' % (tag, desc, trn.shape[0], trn.shape[1], trn.nnz, 2 * trn.nnz, d, L, B),
                       'edges_per_step': edges, 'parallelism': 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0, 'traffic': None,
                         'kernel': type(prof[0][2]).__name__, 'avg_launch_us': float(np.mean(k_ms)) * 1e3, 'launches': len(prof),
                         'launch_timing': 'HIP events around every SpMM launch of the timed region',
                         'algorithmic_bytes_per_launch': float(np.mean(k_bytes))},
            'extras': {'spmm_launches_per_step': len(prof) // steps, 'spmm_ms_per_step': float(np.sum(k_ms)) / steps,
                       'ms_per_step_as_one_hip_graph': graph_ms, **({'hip_graph_error': graph_err} if graph_err else {})}}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    args = ap.parse_args()
    for tag in CONFIGS:
        print(json.dumps(run_config(tag, args.steps)), flush=True)
