"""Pin the CPU oracle (oracle/ref_expr.py) against outputs of the REAL reference
recorded by oracle/make_golden.py (tests/golden/*.npz).  CPU only."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import ref_expr as R

CASES_FULL = [('tiny', 64, 3), ('tiny', 32, 2)]


def _load(golden_dir, case, model, d, L):
    g = np.load(os.path.join(golden_dir, '%s_%s_d%d_L%d.npz' % (case, model, d, L)))
    cfg = json.loads(str(g['cfg']))
    return g, cfg


def _trn(g):
    n_user, n_item = g['shape']
    return sp.coo_matrix((np.ones(len(g['trn_row']), dtype=np.float32), (g['trn_row'], g['trn_col'])),
                         shape=(int(n_user), int(n_item)))


def _params(g, names=('user_embeds', 'item_embeds')):
    return [torch.tensor(g['param_' + n], requires_grad=True) for n in names]


def _batch(g):
    return [torch.from_numpy(g[k]) for k in ('ancs', 'poss', 'negs')]


@pytest.mark.parametrize('case,d,L', CASES_FULL)
def test_adjacency_matches_reference(golden_dir, case, d, L):
    g, _ = _load(golden_dir, case, 'lightgcn', d, L)
    idx, vals, n = R.normalized_bipartite_coo(_trn(g))
    assert np.array_equal(idx, g['adj_idx'])          # same entries in the same (column-major) order
    assert np.array_equal(vals, g['adj_val'])         # bit-identical values
    dense = torch.sparse_coo_tensor(torch.from_numpy(idx), torch.from_numpy(vals), (n, n)).to_dense()
    assert torch.equal(dense, dense.T)                # symmetric


@pytest.mark.parametrize('case,d,L', CASES_FULL)
def test_lightgcn_step(golden_dir, case, d, L):
    torch.set_num_threads(1)
    g, cfg = _load(golden_dir, case, 'lightgcn', d, L)
    idx, vals, n = R.normalized_bipartite_coo(_trn(g))
    adj = R.torch_adj_from(idx, vals, n)
    ue, ie = _params(g)
    assert int(g['n_draws']) == 1 and int(g['n_props']) == L
    mask_draw = torch.from_numpy(g['draw_0'])
    # per-layer outputs of the dropped graph
    _, _, layers = R.lightgcn_forward(adj, ue, ie, L, cfg['keep_rate'], mask_draw, return_layers=True)
    for l in range(L):
        np.testing.assert_allclose(layers[l + 1].detach().numpy(), g['prop_%d' % l], rtol=0, atol=1e-7)
    loss, parts = R.lightgcn_cal_loss(adj, ue, ie, _batch(g), L, cfg['keep_rate'], cfg['reg_weight'], mask_draw)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    np.testing.assert_allclose(parts['bpr_loss'].item(), g['part_bpr_loss'], rtol=1e-6)
    np.testing.assert_allclose(ue.grad.numpy(), g['grad_user_embeds'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(ie.grad.numpy(), g['grad_item_embeds'], rtol=0, atol=1e-7)


@pytest.mark.parametrize('case,d,L', CASES_FULL)
def test_sgl_step(golden_dir, case, d, L):
    torch.set_num_threads(1)
    g, cfg = _load(golden_dir, case, 'sgl', d, L)
    idx, vals, n = R.normalized_bipartite_coo(_trn(g))
    adj = R.torch_adj_from(idx, vals, n)
    ue, ie = _params(g)
    assert int(g['n_draws']) == 2
    draws = (torch.from_numpy(g['draw_0']), torch.from_numpy(g['draw_1']))
    loss, parts = R.sgl_cal_loss(adj, ue, ie, _batch(g), L, cfg['keep_rate'], cfg['reg_weight'], cfg['cl_weight'],
                                 cfg['temperature'], draws)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    np.testing.assert_allclose(parts['cl_loss'].item(), g['part_cl_loss'], rtol=1e-6)
    np.testing.assert_allclose(ue.grad.numpy(), g['grad_user_embeds'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ie.grad.numpy(), g['grad_item_embeds'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('case,d,L', CASES_FULL)
def test_simgcl_step(golden_dir, case, d, L):
    torch.set_num_threads(1)
    g, cfg = _load(golden_dir, case, 'simgcl', d, L)
    idx, vals, n = R.normalized_bipartite_coo(_trn(g))
    adj = R.torch_adj_from(idx, vals, n)
    ue, ie = _params(g)
    assert int(g['n_draws']) == 2 * L       # view 1 all layers, then view 2
    nd1 = [torch.from_numpy(g['draw_%d' % i]) for i in range(L)]
    nd2 = [torch.from_numpy(g['draw_%d' % (L + i)]) for i in range(L)]
    loss, parts = R.simgcl_cal_loss(adj, ue, ie, _batch(g), L, cfg['reg_weight'], cfg['cl_weight'],
                                    cfg['temperature'], cfg['eps'], (nd1, nd2))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    np.testing.assert_allclose(parts['cl_loss'].item(), g['part_cl_loss'], rtol=1e-6)
    np.testing.assert_allclose(ue.grad.numpy(), g['grad_user_embeds'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ie.grad.numpy(), g['grad_item_embeds'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('case,d,L', CASES_FULL + [('tiny', 128, 2)])      # d = 128: BASELINE cfg 5's embedding size
def test_lightgcl_step(golden_dir, case, d, L):
    torch.set_num_threads(1)
    g, cfg = _load(golden_dir, case, 'lightgcl', d, L)
    adj_ui = R.lightgcl_adj(_trn(g))
    assert np.array_equal(adj_ui.indices().numpy(), g['lgcl_adj_idx'])
    assert np.array_equal(adj_ui.values().numpy(), g['lgcl_adj_val'])
    ue, ie = _params(g)
    ws = [torch.tensor(g['param_Ws_%d_W' % i], requires_grad=True) for i in range(L)]
    svd = tuple(torch.from_numpy(g[k]) for k in ('svd_ut', 'svd_vt', 'svd_u_mul_s', 'svd_v_mul_s'))
    e_u, e_i, g_u, g_i = R.lightgcl_forward(adj_ui, ue, ie, *svd, L)
    np.testing.assert_allclose(e_u.detach().numpy(), g['E_u'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(g_i.detach().numpy(), g['G_i'], rtol=0, atol=1e-6)
    loss, parts = R.lightgcl_cal_loss(adj_ui, ue, ie, ws, svd, _batch(g), L, cfg['reg_weight'], cfg['cl_weight'],
                                      cfg['temp'])
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-6)
    np.testing.assert_allclose(parts['cl_loss'].item(), g['part_cl_loss'], rtol=1e-6)
    np.testing.assert_allclose(ue.grad.numpy(), g['grad_user_embeds'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ie.grad.numpy(), g['grad_item_embeds'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ws[0].grad.numpy(), g['grad_Ws_0_W'], rtol=1e-5, atol=1e-9)


def _fill_params(shapes):
    out = []
    for i, shp in enumerate(shapes):
        fill = np.random.default_rng(100 + i).uniform(-0.05, 0.05, size=shp).astype(np.float32)
        out.append(torch.tensor(fill, requires_grad=True))
    return out


def _check_sampled(grad, g, key):
    np.testing.assert_allclose(grad[::997].numpy(), g['gradrows_' + key], rtol=1e-4, atol=1e-7)
    s = np.array([grad.double().sum().item(), grad.double().abs().sum().item()])
    np.testing.assert_allclose(s, g['gradsum_' + key], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('model', ['lightgcn', 'sgl', 'simgcl'])
def test_real_yelp_step(golden_dir, model):
    """Real yelp interactions (stored in the fixture), d=64, L=2, B=4096: the oracle reproduces the
    reference's loss parts and sampled gradient rows.  The tables are a closed-form seeded fill and
    the torch.rand stream is re-seeded right before the step, exactly as make_golden.py did."""
    g, cfg = _load(golden_dir, 'yelp', model, 64, 2)
    trn = _trn(g)
    idx, vals, n = R.normalized_bipartite_coo(trn)
    adj = R.torch_adj_from(idx, vals, n)
    ue, ie = _fill_params([(trn.shape[0], 64), (trn.shape[1], 64)])
    np.testing.assert_array_equal(ue.detach()[::997].numpy(), g['paramrows_user_embeds'])
    torch.manual_seed(2023 + 1)
    L = 2
    if model == 'lightgcn':
        loss, parts = R.lightgcn_cal_loss(adj, ue, ie, _batch(g), L, cfg['keep_rate'], cfg['reg_weight'])
    elif model == 'sgl':
        loss, parts = R.sgl_cal_loss(adj, ue, ie, _batch(g), L, cfg['keep_rate'], cfg['reg_weight'],
                                     cfg['cl_weight'], cfg['temperature'])
    else:
        loss, parts = R.simgcl_cal_loss(adj, ue, ie, _batch(g), L, cfg['reg_weight'], cfg['cl_weight'],
                                        cfg['temperature'], cfg['eps'])
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=2e-6)
    for k, v in parts.items():
        np.testing.assert_allclose(v.item(), g['part_' + k], rtol=2e-6)
    _check_sampled(ue.grad, g, 'user_embeds')
    _check_sampled(ie.grad, g, 'item_embeds')


@pytest.mark.parametrize('model_name', ['lightgcn', 'sgl', 'simgcl'])
def test_training_trajectory_of_the_reference_is_reproduced_on_the_host(model_name):
    """A 2-epoch training RUN of the real reference (24 Adam steps on the tiny dataset; golden traj_*.npz) replayed
    with this repo's host logic (config, data handler, negative sampler, shuffled loader, RNG consumption order) and
    the oracle's loss expressions: same initial parameters, same per-step losses, same final embeddings."""
    from tests import helpers as H
    from sslrec_amd.config.configurator import configs
    g, cfg, opt_cfg, meta = H.load_trajectory(model_name)
    dh = H.trajectory_setup(model_name, g, cfg, opt_cfg, meta, 'cpu')
    torch.set_num_threads(1)
    n_user, n_item = (int(x) for x in g['shape'])
    d = cfg['embedding_size']
    ue = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_user, d)))      # users first, like the reference
    ie = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_item, d)))
    assert np.array_equal(ue.detach().numpy(), g['init_user_embeds'])
    assert np.array_equal(ie.detach().numpy(), g['init_item_embeds'])
    opt = torch.optim.Adam([ue, ie], lr=opt_cfg['lr'], weight_decay=opt_cfg['weight_decay'])
    adj = dh.torch_adj
    losses = []
    for _ in range(meta['epochs']):
        dh.train_dataloader.dataset.sample_negs()
        for tem in dh.train_dataloader:
            batch = [x.long() for x in tem]
            opt.zero_grad()
            if model_name == 'lightgcn':
                loss, _ = R.lightgcn_cal_loss(adj, ue, ie, batch, cfg['layer_num'], cfg['keep_rate'], cfg['reg_weight'])
            elif model_name == 'sgl':
                loss, _ = R.sgl_cal_loss(adj, ue, ie, batch, cfg['layer_num'], cfg['keep_rate'], cfg['reg_weight'],
                                         cfg['cl_weight'], cfg['temperature'])
            else:
                loss, _ = R.simgcl_cal_loss(adj, ue, ie, batch, cfg['layer_num'], cfg['reg_weight'], cfg['cl_weight'],
                                            cfg['temperature'], cfg['eps'])
            loss.backward()
            opt.step()
            losses.append(loss.item())
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-6)
    np.testing.assert_allclose(ue.detach().numpy(), g['final_user_embeds'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ie.detach().numpy(), g['final_item_embeds'], rtol=0, atol=2e-6)


def test_training_trajectory_on_real_yelp_is_reproduced_on_the_host():
    """The same pin on the REAL yelp interactions (BASELINE cfg 4 data: 42,712 x 26,822, 182,357 interactions, d=64, L=2,
    B=4096): 2 epochs = 90 Adam steps of the reference's LightGCN (edge drop 0.5) replayed with this repo's host logic and
    the oracle's loss; the golden keeps every 97th row of the initial / final tables and checksums."""
    from tests import helpers as H
    g, cfg, opt_cfg, meta = H.load_trajectory('lightgcn', 64, 2, case='yelp')
    dh = H.trajectory_setup('lightgcn', g, cfg, opt_cfg, meta, 'cpu')
    n_user, n_item = (int(x) for x in g['shape'])
    d = cfg['embedding_size']
    ue = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_user, d)))
    ie = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_item, d)))
    assert np.array_equal(ue.detach().numpy()[::97], g['initrows_user_embeds'])
    assert np.array_equal(ie.detach().numpy()[::97], g['initrows_item_embeds'])
    opt = torch.optim.Adam([ue, ie], lr=opt_cfg['lr'], weight_decay=opt_cfg['weight_decay'])
    losses = []
    for _ in range(meta['epochs']):
        dh.train_dataloader.dataset.sample_negs()
        for tem in dh.train_dataloader:
            batch = [x.long() for x in tem]
            opt.zero_grad()
            loss, _ = R.lightgcn_cal_loss(dh.torch_adj, ue, ie, batch, cfg['layer_num'], cfg['keep_rate'], cfg['reg_weight'])
            loss.backward()
            opt.step()
            losses.append(loss.item())
    np.testing.assert_allclose(losses, g['losses'], rtol=1e-5)
    np.testing.assert_allclose(ue.detach().numpy()[::97], g['finalrows_user_embeds'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(ie.detach().numpy()[::97], g['finalrows_item_embeds'], rtol=0, atol=1e-5)
