"""DataHandlerGeneralCF: loads the three interaction matrices, builds the normalized bipartite
adjacency `torch_adj` and the three DataLoaders -- the contract of the reference's
data_utils/data_handler_general_cf.py:11-95 (attributes trn_file/val_file/tst_file, trn_mat,
torch_adj, train/valid/test_dataloader; sets configs['data']['user_num'/'item_num']).

`torch_adj` is bit-identical to the reference's (same entries, same column-major entry
order, same fp64->fp32 values; pinned by tests/test_host_logic.py against the golden
vectors) but is built directly from the degrees instead of through three sparse-sparse
products, and the CSR work list the HIP SpMM consumes is attached to it at construction so
no kernel ever has to coalesce it (sslrec_amd/graph.py).

Besides the reference's pickle directories (configs['data']['name'] in yelp|gowalla|amazon,
cwd-relative `./datasets/general_cf/sparse_*/`), `configs['data']['synthetic']` = one of
sslrec_amd.data_utils.synth.SHAPES generates a seeded graph of that shape in memory -- the
GPU box has no datasets.
"""
import pickle

import numpy as np
import scipy.sparse as sp
import torch as t
import torch.utils.data as data

from ..config.configurator import configs
from .datasets_general_cf import AllRankTstData, ExactPairwiseLoader, FastPairwiseLoader, PairwiseTrnData, PairwiseWEpochFlagTrnData
from . import synth


class DataHandlerGeneralCF:
    def __init__(self):
        name = configs['data']['name']
        known = {'yelp': 'sparse_yelp', 'gowalla': 'sparse_gowalla', 'amazon': 'sparse_amazon'}
        self.synthetic = configs['data'].get('synthetic')
        if self.synthetic is None and name not in known:
            raise ValueError("unknown general_cf dataset '%s' (yelp|gowalla|amazon, or set data.synthetic)" % name)
        predir = './datasets/general_cf/%s/' % known.get(name, 'synthetic')
        self.trn_file = predir + 'train_mat.pkl'
        self.val_file = predir + 'valid_mat.pkl'
        self.tst_file = predir + 'test_mat.pkl'

    def _load_one_mat(self, file):
        """pickle -> binarized float32 scipy COO (reference :22-35)."""
        if self.synthetic is not None:
            mat = self._synthetic_mats()[file]
        else:
            with open(file, 'rb') as fs:
                mat = pickle.load(fs)
        mat = (mat != 0).astype(np.float32)
        if type(mat) != sp.coo_matrix:
            mat = sp.coo_matrix(mat)
        return mat

    def _synthetic_mats(self):
        if not hasattr(self, '_synth_cache'):
            seed = configs['data'].get('synthetic_seed', 2023)
            trn = synth.make_dataset(self.synthetic, seed)
            val = synth.split_holdout(trn, configs['data'].get('synthetic_valid_frac', 0.002), seed + 1)
            tst = synth.split_holdout(trn, configs['data'].get('synthetic_test_frac', 0.002), seed + 2)
            self._synth_cache = {self.trn_file: trn, self.val_file: val, self.tst_file: tst}
        return self._synth_cache

    def _normalize_adj(self, rows, cols, n):
        """Symmetric normalization D^-1/2 A D^-1/2 of a binary pattern given as (rows, cols):
        fp64 degrees + 1e-10, d^-1/2 with infinities zeroed (reference :37-51).  Returns the
        float64 value of every entry."""
        degree = np.bincount(rows, minlength=n).astype(np.float64) + 1e-10
        d_inv_sqrt = np.power(degree, -0.5)
        d_inv_sqrt[np.isinf(d_inv_sqrt)] = 0.0
        return d_inv_sqrt[rows] * d_inv_sqrt[cols]

    def _make_torch_adj(self, mat):
        """(U+I)^2 normalized bipartite adjacency as an UNCOALESCED torch sparse COO tensor whose
        entries are ordered by (column, row) -- exactly what the reference's scipy pipeline emits
        (reference :53-73) -- on configs['device']."""
        n_user, n_item = configs['data']['user_num'], configs['data']['item_num']
        n = n_user + n_item
        keys = np.unique(mat.row.astype(np.int64) * n_item + mat.col.astype(np.int64))   # binarize: drop duplicates
        u, i = keys // n_item, keys % n_item + n_user
        rows = np.concatenate([u, i])
        cols = np.concatenate([i, u])
        order = np.lexsort((rows, cols))
        rows, cols = rows[order], cols[order]
        vals = self._normalize_adj(rows, cols, n).astype(np.float32)
        idxs = t.from_numpy(np.vstack([rows, cols]).astype(np.int64))
        adj = t.sparse_coo_tensor(idxs, t.from_numpy(vals), (n, n), check_invariants=False)
        return adj.to(configs['device'])

    def load_data(self):
        trn_mat = self._load_one_mat(self.trn_file)
        tst_mat = self._load_one_mat(self.tst_file)
        val_mat = self._load_one_mat(self.val_file)

        self.trn_mat = trn_mat
        configs['data']['user_num'], configs['data']['item_num'] = trn_mat.shape
        self.torch_adj = self._make_torch_adj(trn_mat)

        if configs['train']['loss'] == 'pairwise':
            trn_data = PairwiseTrnData(trn_mat)
        elif configs['train']['loss'] == 'pairwise_with_epoch_flag':
            trn_data = PairwiseWEpochFlagTrnData(trn_mat)
        else:
            raise NotImplementedError("train.loss '%s'" % configs['train']['loss'])
        val_data = AllRankTstData(val_mat, trn_mat)
        tst_data = AllRankTstData(tst_mat, trn_mat)
        self.valid_dataloader = data.DataLoader(val_data, batch_size=configs['test']['batch_size'], shuffle=False, num_workers=0)
        self.test_dataloader = data.DataLoader(tst_data, batch_size=configs['test']['batch_size'], shuffle=False, num_workers=0)
        dev = configs['device'] if str(configs['device']).startswith('cuda') else None
        if configs['train'].get('fast_loader') and configs['train']['loss'] == 'pairwise':
            self.train_dataloader = FastPairwiseLoader(trn_data, configs['train']['batch_size'], device=dev)
        elif configs['train']['loss'] == 'pairwise' and not configs['train'].get('torch_dataloader'):
            # the reference's DataLoader(shuffle=True) batch for batch and draw for draw, as array slices (ExactPairwiseLoader)
            self.train_dataloader = ExactPairwiseLoader(trn_data, configs['train']['batch_size'], device=dev)
        else:       # `train.torch_dataloader: true`, or a dataset with per-sample side effects (the epoch flag): the reference's own loader
            self.train_dataloader = data.DataLoader(trn_data, batch_size=configs['train']['batch_size'], shuffle=True, num_workers=0)
