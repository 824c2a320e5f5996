"""Mint golden vectors by running the REAL reference (HKUDS/SSLRec, read-only at
/root/reference) on the CPU.  TEST INFRASTRUCTURE ONLY; runs in the authoring
container (the GPU box has no /root/reference) and writes small fixtures to
tests/golden/.  Re-run:  python oracle/make_golden.py

What it does (recipe from SURVEY.md §8c / Appendix B):
  * builds a scratch cwd with `config -> /root/reference/config` and a
    `datasets/general_cf/sparse_yelp/` directory holding either a tiny seeded
    synthetic dataset (case "tiny") or symlinks to the real yelp pickles (case
    "yelp"); the reference's data handler only knows yelp|gowalla|amazon;
  * in a fresh subprocess per model (the reference parses argv and builds a
    global `configs` at import time) imports the reference, overrides
    d / L / batch size the way trainer/tuner.py does, seeds numpy+torch,
    builds data handler + model, samples negatives, takes the first batch and
    runs `cal_loss` + `backward`;
  * records every `torch.rand` draw (edge-drop masks, SimGCL noise), every
    `_propagate` output, the losses and the parameter gradients.

Nothing from the reference is copied; it is only imported and executed.
"""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLD = os.path.join(REPO, 'tests', 'golden')

WORKER = r'''
import sys, json, pickle
import numpy as np
model_name, out_path, d, L, B, seed, full = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
sys.argv = ['x', '--model', model_name, '--dataset', 'yelp', '--device', 'cpu']
sys.path.insert(0, '/root/reference')
import torch
torch.Tensor.cuda = lambda self, *a, **k: self       # aug_utils.py:130, lightgcl.py:22,63 hard-code .cuda()
torch.set_num_threads(1)                              # deterministic reduction order for the fixture
draws = []
_orig_rand = torch.rand
def _rec_rand(*a, **k):
    r = _orig_rand(*a, **k)
    draws.append(r.clone())
    return r
torch.rand = _rec_rand
from config.configurator import configs
configs['model']['embedding_size'] = d
configs['model']['layer_num'] = L
configs['train']['batch_size'] = B
from data_utils.build_data_handler import build_data_handler
from models.bulid_model import build_model
torch.manual_seed(seed); np.random.seed(seed)
dh = build_data_handler(); dh.load_data()
model = build_model(dh)
props = []
if hasattr(model, '_propagate'):
    _orig_prop = model._propagate
    def _rec_prop(adj, embeds):
        y = _orig_prop(adj, embeds)
        props.append(y.detach().clone())
        return y
    model._propagate = _rec_prop
dh.train_dataloader.dataset.sample_negs()
batch = [x.long() for x in next(iter(dh.train_dataloader))]
if not full:
    # big real-data case: the tables are too large to commit, so overwrite them with a
    # closed-form seeded fill the test can regenerate (numpy Generator streams are stable)
    with torch.no_grad():
        for i, (name, p) in enumerate(model.named_parameters()):
            fill = np.random.default_rng(100 + i).uniform(-0.05, 0.05, size=tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(fill))
# re-seed right before the step so a test can reproduce the torch.rand stream
torch.manual_seed(seed + 1)
n_draws_before = len(draws)
loss, parts = model.cal_loss(batch)
loss.backward()
out = {}
trn = dh.trn_mat
out['trn_row'] = trn.row.astype(np.int32); out['trn_col'] = trn.col.astype(np.int32)
out['shape'] = np.array(trn.shape, dtype=np.int64)
out['cfg'] = np.array(json.dumps({k: configs['model'][k] for k in configs['model']}))
out['ancs'] = batch[0].numpy(); out['poss'] = batch[1].numpy(); out['negs'] = batch[2].numpy()
out['loss'] = loss.detach().numpy()
for k, v in parts.items():
    out['part_' + k] = v.detach().numpy() if hasattr(v, 'detach') else np.array(v)
for name, p in model.named_parameters():
    key = name.replace('.', '_')
    if full or p.numel() <= 1 << 16:
        out['param_' + key] = p.detach().numpy()
        out['grad_' + key] = p.grad.numpy()
    else:   # big real-data case: keep a strided sample + checksums
        out['gradsum_' + key] = np.array([p.grad.double().sum().item(), p.grad.double().abs().sum().item()])
        out['gradrows_' + key] = p.grad[::997].numpy()
        out['paramrows_' + key] = p.detach()[::997].numpy()
if full:
    adj = dh.torch_adj
    out['adj_idx'] = adj._indices().numpy(); out['adj_val'] = adj._values().numpy()
    for i, r in enumerate(draws[n_draws_before:]):
        out['draw_%d' % i] = r.numpy()
    for i, y in enumerate(props):
        out['prop_%d' % i] = y.numpy()
    if model_name == 'lightgcl':
        out['lgcl_adj_idx'] = model.adj.indices().numpy(); out['lgcl_adj_val'] = model.adj.values().numpy()
        out['E_u'] = model.E_u.detach().numpy(); out['E_i'] = model.E_i.detach().numpy()
        out['G_u'] = model.G_u.detach().numpy(); out['G_i'] = model.G_i.detach().numpy()
if model_name == 'lightgcl':
    out['svd_ut'] = model.ut.numpy(); out['svd_vt'] = model.vt.numpy()
    out['svd_u_mul_s'] = model.u_mul_s.numpy(); out['svd_v_mul_s'] = model.v_mul_s.numpy()
out['n_draws'] = np.array(len(draws) - n_draws_before)
out['n_props'] = np.array(len(props))
np.savez_compressed(out_path, **out)
print(model_name, 'loss', float(loss), {k: float(v) for k, v in parts.items()})
'''


# A short TRAINING RUN of the reference: init_seed -> data handler -> model -> Adam exactly as trainer/trainer.py does
# (sample_negs + shuffled DataLoader per epoch; zero_grad, cal_loss, backward, step per batch), recording the initial
# parameters, every step's loss and the final parameters.
TRAJ_WORKER = r'''
import sys, json
import numpy as np
model_name, out_path, d, L, B, seed, epochs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
full = int(sys.argv[8]) if len(sys.argv) > 8 else 1
sys.argv = ['x', '--model', model_name, '--dataset', 'yelp', '--device', 'cpu']
sys.path.insert(0, '/root/reference')
import torch
torch.Tensor.cuda = lambda self, *a, **k: self
torch.set_num_threads(1)
from config.configurator import configs
configs['model']['embedding_size'] = d
configs['model']['layer_num'] = L
configs['train']['batch_size'] = B
from data_utils.build_data_handler import build_data_handler
from models.bulid_model import build_model
torch.manual_seed(seed); np.random.seed(seed)                      # trainer.init_seed (trainer/trainer.py:26-36)
dh = build_data_handler(); dh.load_data()
model = build_model(dh)
init = {n: p.detach().clone().numpy() for n, p in model.named_parameters()}
opt_cfg = configs['optimizer']
opt = torch.optim.Adam(model.parameters(), lr=opt_cfg['lr'], weight_decay=opt_cfg['weight_decay'])   # trainer.py:45-49
losses = []
for ep in range(epochs):                                             # trainer.py:51-72
    dh.train_dataloader.dataset.sample_negs()
    for tem in dh.train_dataloader:
        batch = [x.long() for x in tem]
        opt.zero_grad()
        loss, parts = model.cal_loss(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
out = {}
trn = dh.trn_mat
out['trn_row'] = trn.row.astype(np.int32); out['trn_col'] = trn.col.astype(np.int32)
out['shape'] = np.array(trn.shape, dtype=np.int64)
out['cfg'] = np.array(json.dumps({k: configs['model'][k] for k in configs['model']}))
out['opt'] = np.array(json.dumps({'lr': opt_cfg['lr'], 'weight_decay': opt_cfg['weight_decay']}))
out['meta'] = np.array(json.dumps({'seed': seed, 'epochs': epochs, 'batch_size': B}))
out['losses'] = np.array(losses, dtype=np.float64)
for n, p in model.named_parameters():
    key = n.replace('.', '_')
    if full:
        out['init_' + key] = init[n]
        out['final_' + key] = p.detach().numpy()
    else:       # real-data case: every 97th row + checksums (the initial parameters are re-derived from the seed)
        out['initrows_' + key] = init[n][::97]
        out['finalrows_' + key] = p.detach().numpy()[::97]
        out['finalsum_' + key] = np.array([p.detach().double().sum().item(), p.detach().double().abs().sum().item()])
np.savez_compressed(out_path, **out)
print(model_name, 'steps', len(losses), 'first/last loss', losses[0], losses[-1])
'''


# The reference's BATCH STREAM alone: `sample_negs()` + the shuffled DataLoader for a few epochs (trainer/trainer.py:51-62,
# data_utils/datasets_general_cf.py:13-20, data_handler_general_cf.py:95), recording what a step receives and where the two
# generators (numpy global, torch CPU) stand afterwards.  Pins the native sampler and the array-slicing loader.
BATCH_WORKER = r'''
import sys, json, hashlib
import numpy as np
out_path, B, seed, epochs, full = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sys.argv = ['x', '--model', 'lightgcn', '--dataset', 'yelp', '--device', 'cpu']
sys.path.insert(0, '/root/reference')
import torch
from config.configurator import configs
configs['train']['batch_size'] = B
from data_utils.build_data_handler import build_data_handler
torch.manual_seed(seed); np.random.seed(seed)
dh = build_data_handler(); dh.load_data()
out = {}
trn = dh.trn_mat
out['trn_row'] = trn.row.astype(np.int32); out['trn_col'] = trn.col.astype(np.int32)
out['shape'] = np.array(trn.shape, dtype=np.int64)
out['meta'] = np.array(json.dumps({'seed': seed, 'epochs': epochs, 'batch_size': B}))
for ep in range(epochs):
    dh.train_dataloader.dataset.sample_negs()
    h = hashlib.sha256()
    h.update(dh.train_dataloader.dataset.negs.astype(np.int32).tobytes())
    out['negs_sha_%d' % ep] = np.array(h.hexdigest())
    if full:
        out['negs_%d' % ep] = dh.train_dataloader.dataset.negs.astype(np.int32).copy()
    h = hashlib.sha256()
    batches = []
    for tem in dh.train_dataloader:
        assert all(x.dtype == torch.int32 for x in tem)
        arr = np.stack([x.numpy() for x in tem])            # [3, b] int32
        h.update(arr.tobytes())
        batches.append(arr)
    out['batches_sha_%d' % ep] = np.array(h.hexdigest())
    out['n_batches_%d' % ep] = np.array(len(batches))
    out['first_batch_%d' % ep] = batches[0]; out['last_batch_%d' % ep] = batches[-1]
    if full:
        for i, a in enumerate(batches):
            out['batch_%d_%d' % (ep, i)] = a
st = np.random.get_state()
out['np_pos'] = np.array(st[2]); out['np_key_sha'] = np.array(hashlib.sha256(st[1].astype(np.uint32).tobytes()).hexdigest())
out['np_next'] = np.array([np.random.randint(1 << 30) for _ in range(4)])
out['torch_state_sha'] = np.array(hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest())
out['torch_next'] = torch.rand(4).numpy()
np.savez_compressed(out_path, **out)
print('batches', out_path, [int(out['n_batches_%d' % e]) for e in range(epochs)])
'''


def run_batches(case, B, seed, epochs):
    root = _scratch(case)
    with open(os.path.join(root, 'batch_worker.py'), 'w') as fs:
        fs.write(BATCH_WORKER)
    out = os.path.join(GOLD, 'batches_%s_B%d.npz' % (case, B))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    subprocess.run([sys.executable, 'batch_worker.py', out, str(B), str(seed), str(epochs), str(int(case == 'tiny'))],
                   cwd=root, env=env, check=True)
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')


def run_trajectory(case, d, L, B, seed, epochs, models=('lightgcn', 'sgl', 'simgcl')):
    root = _scratch(case)
    with open(os.path.join(root, 'traj_worker.py'), 'w') as fs:
        fs.write(TRAJ_WORKER)
    for model in models:
        out = os.path.join(GOLD, 'traj_%s_%s_d%d_L%d.npz' % (case, model, d, L))
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        subprocess.run([sys.executable, 'traj_worker.py', model, out, str(d), str(L), str(B), str(seed), str(epochs),
                        str(int(case == 'tiny'))], cwd=root, env=env, check=True)
        print('wrote', out, os.path.getsize(out) // 1024, 'KiB')


def _scratch(case):
    sys.path.insert(0, REPO)
    from sslrec_amd.data_utils.synth import make_dataset, split_holdout
    root = tempfile.mkdtemp(prefix='sslrec_golden_')
    os.symlink(os.path.join(REF, 'config'), os.path.join(root, 'config'))
    ddir = os.path.join(root, 'datasets', 'general_cf', 'sparse_yelp')
    os.makedirs(ddir)
    if case == 'yelp':
        for f in ('train_mat.pkl', 'valid_mat.pkl', 'test_mat.pkl'):
            os.symlink(os.path.join(REF, 'datasets/general_cf/sparse_yelp', f), os.path.join(ddir, f))
    else:
        trn = make_dataset('tiny', seed=2023)
        val = split_holdout(trn, 0.05, 1)
        tst = split_holdout(trn, 0.2, 2)
        for f, m in (('train_mat.pkl', trn), ('valid_mat.pkl', val), ('test_mat.pkl', tst)):
            with open(os.path.join(ddir, f), 'wb') as fs:
                pickle.dump(m, fs)
    with open(os.path.join(root, 'worker.py'), 'w') as fs:
        fs.write(WORKER)
    return root


def run_case(case, d, L, B, seed, full, models=('lightgcn', 'sgl', 'simgcl', 'lightgcl')):
    root = _scratch(case)
    os.makedirs(GOLD, exist_ok=True)
    for model in models:
        out = os.path.join(GOLD, '%s_%s_d%d_L%d.npz' % (case, model, d, L))
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        subprocess.run([sys.executable, 'worker.py', model, out, str(d), str(L), str(B), str(seed), str(int(full))],
                       cwd=root, env=env, check=True)
        print('wrote', out, os.path.getsize(out) // 1024, 'KiB')


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference tree not present; golden vectors can only be minted in the authoring container')
    run_case('tiny', d=64, L=3, B=256, seed=2023, full=True)
    run_case('tiny', d=32, L=2, B=256, seed=7, full=True)
    run_case('yelp', d=64, L=2, B=4096, seed=2023, full=False)
    run_case('tiny', d=128, L=2, B=256, seed=11, full=True, models=('lightgcl',))      # BASELINE cfg 5's embedding size
    run_trajectory('tiny', d=64, L=3, B=256, seed=2023, epochs=2)
    run_trajectory('yelp', d=64, L=2, B=4096, seed=2023, epochs=2, models=('lightgcn', 'sgl'))
    run_batches('tiny', B=256, seed=2023, epochs=3)
    run_batches('yelp', B=4096, seed=2023, epochs=2)
