"""The default batch stream -- native negative sampler (csrc/sampler.cpp) + array-slicing loader (ExactPairwiseLoader) -- against
the REAL reference's `sample_negs()` + `DataLoader(shuffle=True)` (goldens batches_*.npz, minted by oracle/make_golden.py:
run_batches): every negative, every batch, and the state of BOTH generators afterwards, bit for bit.  CPU only: the sampler is
host code of the C-ABI library."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from sslrec_amd.config.configurator import configs, load_config
from tests import helpers as H


def _load(case, B):
    g = np.load(os.path.join(H.GOLDEN, 'batches_%s_B%d.npz' % (case, B)))
    return g, json.loads(str(g['meta']))


def _handler(g, meta, train_over=None):
    over = {'train': dict({'batch_size': meta['batch_size']}, **(train_over or {}))}
    load_config('lightgcn', device='cpu', overrides=over)
    torch.manual_seed(meta['seed'])
    np.random.seed(meta['seed'])
    dh = H.FixtureHandler(H.golden_trn(g))
    dh.load_data()
    return dh


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('case,B', [('tiny', 256), ('yelp', 4096)])
def test_default_batch_stream_is_the_references(case, B):
    from sslrec_amd.data_utils.datasets_general_cf import ExactPairwiseLoader
    g, meta = _load(case, B)
    dh = _handler(g, meta)
    assert isinstance(dh.train_dataloader, ExactPairwiseLoader)            # the default, not an opt-in
    ds = dh.train_dataloader.dataset
    assert np.array_equal(ds.rows, g['trn_row']) and np.array_equal(ds.cols, g['trn_col'])
    for ep in range(meta['epochs']):
        ds.sample_negs()
        assert ds.negs.dtype == np.int32 and _sha(ds.negs) == str(g['negs_sha_%d' % ep])
        if 'negs_%d' % ep in g:
            assert np.array_equal(ds.negs, g['negs_%d' % ep])
        h = hashlib.sha256()
        batches = []
        for tem in dh.train_dataloader:
            assert isinstance(tem, list) and len(tem) == 3 and all(x.dtype == torch.int32 for x in tem)
            arr = np.stack([x.numpy() for x in tem])
            h.update(arr.tobytes())
            batches.append(arr)
        assert len(batches) == int(g['n_batches_%d' % ep]) == len(dh.train_dataloader)
        assert np.array_equal(batches[0], g['first_batch_%d' % ep]) and np.array_equal(batches[-1], g['last_batch_%d' % ep])
        assert h.hexdigest() == str(g['batches_sha_%d' % ep])
        i = 0
        while 'batch_%d_%d' % (ep, i) in g:
            assert np.array_equal(batches[i], g['batch_%d_%d' % (ep, i)])
            i += 1
    st = np.random.get_state()
    assert int(st[2]) == int(g['np_pos']) and _sha(st[1].astype(np.uint32)) == str(g['np_key_sha'])
    assert [np.random.randint(1 << 30) for _ in range(4)] == g['np_next'].tolist()
    assert _sha(torch.get_rng_state().numpy()) == str(g['torch_state_sha'])
    assert np.array_equal(torch.rand(4).numpy(), g['torch_next'])


def test_native_sampler_equals_the_python_loop_and_the_torch_dataloader():
    """the two statements the defaults replace, still in the tree behind `train.python_neg_sampling` / `train.torch_dataloader`"""
    g, meta = _load('tiny', 256)
    dh_ref = _handler(g, meta, {'python_neg_sampling': True, 'torch_dataloader': True})
    assert isinstance(dh_ref.train_dataloader, torch.utils.data.DataLoader)
    ref = []
    for _ in range(2):
        dh_ref.train_dataloader.dataset.sample_negs()
        ref += [np.stack([x.numpy() for x in tem]) for tem in dh_ref.train_dataloader]
    ref_state = (np.random.get_state(), torch.get_rng_state())
    dh = _handler(g, meta)
    got = []
    for _ in range(2):
        dh.train_dataloader.dataset.sample_negs()
        got += [np.stack([x.numpy() for x in tem]) for tem in dh.train_dataloader]
    assert len(ref) == len(got) and all(np.array_equal(a, b) for a, b in zip(ref, got))
    st = np.random.get_state()
    assert st[2] == ref_state[0][2] and np.array_equal(st[1], ref_state[0][1])
    assert torch.equal(torch.get_rng_state(), ref_state[1])


def test_loader_draws_at_the_same_two_moments_as_a_dataloader():
    """iter() takes the base seed at once, the sampler's seed comes with the first batch: a draw between the two lands in the same
    place of the stream as it would with torch's DataLoader"""
    g, meta = _load('tiny', 256)
    outs = []
    for over in ({'torch_dataloader': True}, {}):
        dh = _handler(g, meta, over)
        dh.train_dataloader.dataset.sample_negs()
        it = iter(dh.train_dataloader)
        between = torch.rand(2)
        first = next(it)
        outs.append((between, [x.clone() for x in first]))
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


def test_sampler_edge_cases():
    import ctypes as C
    import scipy.sparse as sp
    from sslrec_amd import _lib
    from sslrec_amd.data_utils.datasets_general_cf import PairwiseTrnData
    load_config('lightgcn', device='cpu')
    # one item short of a full row: the only free item must come out, through many rejections, with numpy's stream
    n_item = 37
    rows = np.repeat(np.arange(3), n_item - 1).astype(np.int32)
    cols = np.concatenate([np.delete(np.arange(n_item), k) for k in (0, 17, 36)]).astype(np.int32)
    mat = sp.coo_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(3, n_item))
    configs['data']['user_num'], configs['data']['item_num'] = mat.shape
    ds = PairwiseTrnData(mat)
    np.random.seed(1)
    ds._sample_negs_python()
    ref, ref_state = ds.negs.copy(), np.random.get_state()
    np.random.seed(1)
    ds.sample_negs()
    assert np.array_equal(ds.negs, ref) and set(ds.negs[rows == 1]) == {17}
    st = np.random.get_state()
    assert st[2] == ref_state[2] and np.array_equal(st[1], ref_state[1])
    # a power-of-two item count (mask == range: no masked rejection) and a single item nobody interacted with (no draw at all)
    for n_item, users in ((64, [0, 1, 1, 0]), (1, [0, 0])):
        mat = sp.coo_matrix((np.ones(1, np.float32), ([1], [0])), shape=(2, n_item))
        configs['data']['user_num'], configs['data']['item_num'] = mat.shape
        ds = PairwiseTrnData(mat)
        ds.rows = np.array(users, dtype=np.int32)
        ds.cols = np.zeros(len(users), dtype=np.int32)
        ds.negs = np.zeros(len(users), dtype=np.int32)
        np.random.seed(3)
        ds._sample_negs_python()
        ref, ref_state = ds.negs.copy(), np.random.get_state()
        np.random.seed(3)
        if hasattr(ds, '_trn_csr'):
            del ds._trn_csr
        ds._trn_csr = (np.array([0, 0, 1], dtype=np.int64), np.array([0], dtype=np.int32))
        ds._rows_i32 = ds.rows
        ds.sample_negs()
        assert np.array_equal(ds.negs, ref)
        st = np.random.get_state()
        assert st[2] == ref_state[2] and np.array_equal(st[1], ref_state[1])
    # a user who interacted with every item: the reference never returns; the library refuses
    lib = _lib.load()
    key = np.zeros(624, dtype=np.uint32)
    pos = C.c_int32(624)
    rowptr, col = np.array([0, 2], dtype=np.int64), np.array([0, 1], dtype=np.int32)
    users, negs = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
    rc = lib.sslrec_sample_negs_mt19937(key.ctypes.data, C.addressof(pos), users.ctypes.data, 1, rowptr.ctypes.data, col.ctypes.data,
                                        1, 2, negs.ctypes.data, None)
    assert rc == _lib.E_BADARG
