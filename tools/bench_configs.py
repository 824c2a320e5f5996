#!/usr/bin/env python
"""One JSON line per BASELINE.json config that fits one GPU (the headline cfg 2 is bench.py's own line):
  cfg1-shape  LightGCN step on the gowalla-shaped synthetic graph, d=32, L=2
  cfg3        SimGCL step on the amazon-book-shaped graph, d=64, L=3 (perf mode: device RNG)
  cfg4        SGL-ED step on the REAL yelp interactions (tests/golden/yelp_lightgcn_d64_L2.npz), d=64, L=3, keep 0.5
Each line: step time, propagated (kept) directed edges per second, and the HBM roofline of the SpMM launches inside the
step (HIP events on the launch stream, algorithmic bytes of SURVEY.md §8d) -- same conventions as bench.py.
usage: python tools/bench_configs.py [--steps 30] > profiles/rNN/configs.jsonl"""
import argparse, json, os, sys
import numpy as np, scipy.sparse as sp, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd import ops
from sslrec_amd.config.configurator import configs, load_config
from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
from sslrec_amd.models.bulid_model import build_model
from oracle import ref_expr as R          # only the adjacency normalization of the data handler's host logic

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
args = ap.parse_args()
dev = 'cuda:0'


def yelp_real():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'yelp_lightgcn_d64_L2.npz'))
    U, I = (int(v) for v in z['shape'])
    return sp.coo_matrix((np.ones(z['trn_row'].size, dtype=np.float32), (z['trn_row'], z['trn_col'])), shape=(U, I))


def run(tag, model_name, trn, d, L, extra_model=None, synthetic_name='tiny'):
    over = {'data': {'synthetic': synthetic_name}, 'model': {'embedding_size': d, 'layer_num': L, 'device_rng': True}}
    over['model'].update(extra_model or {})
    load_config(model_name, device=dev, overrides=over)
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
    for _ in range(5):
        step()
    ops.PROFILE = []
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    ms = e0.elapsed_time(e1) / args.steps
    k_ms = [a.elapsed_time(b) for a, b, *_ in prof]
    k_bytes = [lay.algorithmic_bytes(dd, acc=acc, write_y=wy) - (1.0 - frac) * lay.nnz * 8 for _, _, lay, dd, acc, wy, frac in prof]
    ach = float(np.sum(k_bytes)) / (float(np.sum(k_ms)) * 1e-3) / 1e9
    n_launch = len(prof) // args.steps
    line = {'config': tag, 'model': model_name, 'graph': '%dx%d, %d interactions (%d directed entries)' % (trn.shape[0], trn.shape[1], trn.nnz, 2 * trn.nnz),
            'd': d, 'L': L, 'B': B, 'rng': 'device (Philox in the kernels)', 'ms_per_step': ms,
            'spmm_launches_per_step': n_launch, 'spmm_ms_per_step': float(np.sum(k_ms)) / args.steps,
            'spmm_kernel': type(prof[0][2]).__name__, 'spmm_avg_launch_us': float(np.mean(k_ms)) * 1e3,
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0,
                         'algorithmic_bytes_per_launch': float(np.mean(k_bytes))}}
    print(json.dumps(line), flush=True)


from sslrec_amd.data_utils.synth import make_dataset
run('cfg1-shape (gowalla-shaped synthetic)', 'lightgcn', R.binarize_coo(make_dataset('gowalla')), 32, 2, {'keep_rate': 0.5})
run('cfg3 (amazon-book-shaped synthetic)', 'simgcl', R.binarize_coo(make_dataset('amazon-book')), 64, 3)
run('cfg4 (real yelp interactions)', 'sgl', R.binarize_coo(yelp_real()), 64, 3, {'keep_rate': 0.5})
run('cfg4-shape LightGCN (real yelp interactions, keep 0.5)', 'lightgcn', R.binarize_coo(yelp_real()), 64, 3, {'keep_rate': 0.5})
