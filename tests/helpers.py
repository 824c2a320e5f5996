"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import json
import os

import numpy as np
import scipy.sparse as sp
import torch

from sslrec_amd.config.configurator import configs, load_config
from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(case, model, d, L):
    g = np.load(os.path.join(GOLDEN, '%s_%s_d%d_L%d.npz' % (case, model, d, L)))
    return g, json.loads(str(g['cfg']))


def golden_trn(g):
    n_user, n_item = (int(x) for x in g['shape'])
    return sp.coo_matrix((np.ones(len(g['trn_row']), dtype=np.float64), (g['trn_row'], g['trn_col'])),
                         shape=(n_user, n_item))


class FixtureHandler(DataHandlerGeneralCF):
    """Data handler fed from an in-memory interaction matrix instead of pickles."""

    def __init__(self, trn_mat):
        self._fixture = trn_mat
        self.synthetic = None
        self.trn_file = self.val_file = self.tst_file = '<fixture>'

    def _load_one_mat(self, file):
        mat = (self._fixture != 0).astype(np.float32)
        return sp.coo_matrix(mat)

    def load_adj_only(self):
        self.trn_mat = self._load_one_mat(self.trn_file)
        configs['data']['user_num'], configs['data']['item_num'] = self.trn_mat.shape
        self.torch_adj = self._make_torch_adj(self.trn_mat)
        return self


def setup_model(model_name, g, cfg, device, d, L, extra_model_cfg=None):
    """configs + handler + model with the golden's hyper-parameters and parameter values."""
    from sslrec_amd.models.bulid_model import build_model
    over = {'model': dict(cfg)}
    over['model'].update({'embedding_size': d, 'layer_num': L})
    if extra_model_cfg:
        over['model'].update(extra_model_cfg)
    load_config(model_name, device=device, overrides=over)
    dh = FixtureHandler(golden_trn(g)).load_adj_only()
    model = build_model(dh).to(device)
    return dh, model


def set_params_from_golden(model, g):
    with torch.no_grad():
        for name, p in model.named_parameters():
            key = 'param_' + name.replace('.', '_')
            p.copy_(torch.from_numpy(g[key]).to(p.device))


def set_params_seeded_fill(model):
    """the closed-form fill oracle/make_golden.py used for the big real-data case"""
    with torch.no_grad():
        for i, (name, p) in enumerate(model.named_parameters()):
            fill = np.random.default_rng(100 + i).uniform(-0.05, 0.05, size=tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(fill).to(p.device))


class ReplayRand:
    """Replays recorded torch.rand draws (in order) in place of torch.rand."""

    def __init__(self, draws):
        self.draws = list(draws)
        self.i = 0

    def __call__(self, *size, **kw):
        r = self.draws[self.i]
        self.i += 1
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        assert tuple(r.shape) == shape, (tuple(r.shape), shape)
        dev = kw.get('device')
        return r.clone().to(dev) if dev is not None else r.clone()


def golden_draws(g):
    return [torch.from_numpy(g['draw_%d' % i]) for i in range(int(g['n_draws']))]


def batch_from_golden(g, device):
    return [torch.from_numpy(g[k]).to(device) for k in ('ancs', 'poss', 'negs')]


def walk_packed(lay, x, dtype=np.float64, col=None, val=None, r_len=None, w_len=None):
    """host-side walk of a PackedLayout exactly the way spmm_stream_kernel reads it"""
    G = lay.G
    col = lay.col.numpy() if col is None else col
    val = lay.val.numpy() if val is None else val
    ws = lay.w_start.numpy()
    wl = lay.w_len.numpy() if w_len is None else w_len
    rp = lay.r_ptr.numpy()
    rl = lay.r_len.numpy() if r_len is None else r_len
    rd = lay.r_dst.numpy()
    y = np.full((lay.n_rows, x.shape[1]), np.nan, dtype=dtype)
    part = np.zeros((max(lay.n_slots, 1), x.shape[1]), dtype=dtype)
    for w in range(lay.n_waves):
        load = 0
        for k in range(rp[w], rp[w + 1]):
            acc = np.zeros((G, x.shape[1]), dtype=dtype)          # one partial sum per lane group
            for _ in range(rl[k]):
                for sub in range(G):
                    e = ws[w] + (load >> 2) * 4 * G + sub * 4 + (load & 3)
                    if col[e] >= 0:
                        acc[sub] = acc[sub] + dtype(val[e]) * x[col[e]].astype(dtype)
                load += 1
            tot = acc.sum(0, dtype=dtype)
            if rd[k] >= 0:
                assert np.isnan(y[rd[k]]).all(), 'row written twice'
                y[rd[k]] = tot
            else:
                part[~rd[k]] = tot
        assert load == wl[w], (load, wl[w])
    lr, lp = lay.long_row.numpy(), lay.long_ptr.numpy()
    for i in range(lay.n_long):
        acc = np.zeros(x.shape[1], dtype=dtype)
        for s in range(lp[i], lp[i + 1]):
            acc = acc + part[s]
        y[lr[i]] = acc
    assert not np.isnan(y).any(), 'some row was never written'
    return y


def walk_bundled(lay, x, dtype=np.float64, val=None):
    """host-side walk of a BundledLayout exactly the way spmm_bundle_kernel reads it: every lane group of a bundle owns one
    output row (or one chunk of a long row) and adds its entries in stored order; checks that every row is written once"""
    G, d = lay.G, lay.d
    LPG = 64 // G
    S = min(16, LPG)
    col = lay.col.cpu().numpy()
    val = lay.val.cpu().numpy() if val is None else val
    ws, wp = lay.w_start.cpu().numpy(), lay.w_ptr.cpu().numpy()
    bs, bd = lay.b_steps.cpu().numpy(), lay.b_dst.cpu().numpy()
    NONE = -2 ** 31
    y = np.full((lay.n_rows, x.shape[1]), np.nan, dtype=dtype)
    part = np.full((max(lay.n_slots, 1), x.shape[1]), np.nan, dtype=dtype)
    n_edges = 0
    assert ws.size == lay.n_waves + 1 and wp[0] == 0 and wp[-1] == lay.n_bundles and ws[-1] == lay.n_elem
    for w in range(lay.n_waves):
        at = int(ws[w])
        for b in range(wp[w], wp[w + 1]):
            assert bs[b] % S == 0 and at % 64 == 0
            for g in range(G):
                acc = np.zeros(x.shape[1], dtype=dtype)
                seen_pad = False
                for st in range(int(bs[b])):
                    e = at + (st // S) * 64 + g * LPG + st % S
                    if col[e] < 0:
                        seen_pad = True
                        continue
                    assert not seen_pad, 'an entry behind a pad'
                    acc = acc + dtype(val[e]) * x[col[e]].astype(dtype)
                    n_edges += 1
                dst = int(bd[b * G + g])
                if dst >= 0:
                    assert np.isnan(y[dst]).all(), 'row written twice'
                    y[dst] = acc
                elif dst != NONE:
                    assert np.isnan(part[~dst]).all(), 'partial slot written twice'
                    part[~dst] = acc
                else:
                    assert not acc.any()
            at += int(bs[b]) // S * 64
        assert at == ws[w + 1], (w, at, ws[w + 1])
    assert n_edges == lay.nnz
    lr, lp = lay.long_row.cpu().numpy(), lay.long_ptr.cpu().numpy()
    for i in range(lay.n_long):
        assert np.isnan(y[lr[i]]).all(), 'long row also written by a bundle'
        y[lr[i]] = part[lp[i]:lp[i + 1]].sum(0)
    assert not np.isnan(y).any(), 'some row was never written'
    return y


def walk_swept(lay, x, dtype=np.float64):
    """host-side walk of a SweptLayout exactly the way spmm_swept_kernel reads it; also checks the
    invariant the kernel relies on: a slot is only ever touched by ONE lane group of its block"""
    G, nb = lay.G, lay.n_blocks
    pack, val = lay.pack.cpu().numpy(), lay.val.cpu().numpy()
    ws, wst = lay.w_start.cpu().numpy(), lay.w_steps.cpu().numpy()
    wfp, cfp, frow = lay.wf_ptr.cpu().numpy(), lay.cf_ptr.cpu().numpy(), lay.f_row.cpu().numpy()
    fstart, fn = lay.f_start.cpu().numpy(), lay.f_n.cpu().numpy()
    acc = np.zeros((nb, lay.n_slots, x.shape[1]), dtype=dtype)
    owner = {}
    n_edges = 0
    S, LPG = min(16, getattr(lay, 'width', lay.d) // 4), 64 // G
    copies = max(1, LPG // 16)
    for w in range(nb * 16):
        assert wst[w] % S == 0 and ws[w] % 64 == 0
        for s in range(int(wst[w])):
            for g in range(G):
                e = ws[w] + (s // S) * 64 + g * LPG + s % S     # lane j = s % S of every 16-lane row of lane group g
                pk = int(pack[e])
                for c in range(1, copies):
                    assert int(pack[e + 16 * c]) == pk and val[e + 16 * c] == val[e], 'row copies of an entry differ'
                if pk == -1:
                    continue
                u = pk & 0xFFFFFFFF
                slot, c = u >> 20, u & 0xFFFFF
                assert owner.setdefault((w // 16, slot), (w, g)) == (w, g), 'slot shared by two lane groups'
                acc[w // 16, slot] += dtype(val[e]) * x[c].astype(dtype)
                n_edges += 1
    assert n_edges == lay.nnz
    y = np.full((lay.n_rows, x.shape[1]), np.nan, dtype=dtype)
    # one-slot rows: flushed by the wave that owns the slot, without waiting for anybody else -- so nobody else may touch it
    assert wfp[0] == 0 and wfp[-1] == cfp[0] and cfp[-1] == lay.n_rows
    for w in range(nb * 16):
        for i in range(wfp[w], wfp[w + 1]):
            assert np.isnan(y[frow[i]]).all(), 'row flushed twice'
            assert fn[i] == 1 and fstart[i] < lay.n_slots
            assert owner.get((w // 16, int(fstart[i])), (w, 0))[0] == w, 'a wave flushes a slot another wave accumulates'
            y[frow[i]] = acc[w // 16, fstart[i]]
    # chunked rows: added up by their workgroup behind its barrier
    for b in range(nb):
        for i in range(cfp[b], cfp[b + 1]):
            assert np.isnan(y[frow[i]]).all(), 'row flushed twice'
            assert fn[i] > 1 and fstart[i] + fn[i] <= lay.n_slots
            y[frow[i]] = acc[b, fstart[i]:fstart[i] + fn[i]].sum(0)
    assert not np.isnan(y).any(), 'some row was never flushed'
    return y


def load_trajectory(model, d=64, L=3, case='tiny'):
    """golden short training run of the real reference (oracle/make_golden.py: run_trajectory)"""
    g = np.load(os.path.join(GOLDEN, 'traj_%s_%s_d%d_L%d.npz' % (case, model, d, L)))
    return g, json.loads(str(g['cfg'])), json.loads(str(g['opt'])), json.loads(str(g['meta']))


def trajectory_setup(model_name, g, cfg, opt, meta, device):
    """configs + seeds + data handler in the order of the reference's main.py / trainer.init_seed"""
    over = {'model': dict(cfg), 'train': {'batch_size': meta['batch_size']},
            'optimizer': {'lr': opt['lr'], 'weight_decay': opt['weight_decay']}}
    load_config(model_name, device=device, overrides=over)
    torch.manual_seed(meta['seed'])
    np.random.seed(meta['seed'])
    dh = FixtureHandler(golden_trn(g))
    dh.load_data()
    return dh
