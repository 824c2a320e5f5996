"""Training / evaluation datasets of the general-CF scenario.  Class names, attributes, item
formats and RNG consumption follow the reference's data_utils/datasets_general_cf.py:6-68, so a
fixed seed yields the same batches:

  * PairwiseTrnData           (anchor, positive, negative) triples; `sample_negs()` draws one
                              negative per interaction, in interaction order, from numpy's GLOBAL
                              generator and rejects items the user interacted with (:13-20);
  * PairwiseWEpochFlagTrnData NCL's variant with an epoch flag (:28-44);
  * AllRankTstData            (user, dense float64 train-mask row) per test user (:46-68).
"""
import numpy as np
import torch.utils.data as data

from ..config.configurator import configs


class PairwiseTrnData(data.Dataset):
    def __init__(self, coomat):
        self.rows, self.cols = coomat.row, coomat.col
        self._coomat = coomat
        self._dokmat = None
        self.negs = np.zeros(len(self.rows)).astype(np.int32)

    @property
    def dokmat(self):
        """the reference's O(1) membership structure (datasets_general_cf.py:10), built on first use: only the Python form of the
        sampler reads it (the native sampler tests membership in the sorted train rows), and `coomat.todok()` of 2.4 M
        interactions takes tens of seconds"""
        if self._dokmat is None:
            self._dokmat = self._coomat.todok()
        return self._dokmat

    def sample_negs(self):
        if configs['train'].get('device_sampler') and configs['train'].get('fast_loader'):
            self.negs_on_device = True          # drawn by FastPairwiseLoader on the device, see sample_negs_device
            return
        if configs['train'].get('fast_neg_sampling'):
            return self._sample_negs_vectorized()
        if configs['train'].get('python_neg_sampling'):
            return self._sample_negs_python()
        self._sample_negs_native()

    def _sample_negs_python(self):
        """the reference's loop itself (datasets_general_cf.py:13-20), one Python iteration per interaction: 2.2 us each.
        Kept as the statement the native sampler is tested against (`train.python_neg_sampling: true`)."""
        n_item = configs['data']['item_num']
        interacted = self.dokmat
        draw = np.random.randint
        for pos, user in enumerate(self.rows):
            candidate = draw(n_item)
            while (user, candidate) in interacted:
                candidate = draw(n_item)
            self.negs[pos] = candidate

    def _sample_negs_native(self):
        """DEFAULT: the same loop in C++ on numpy's own generator state (csrc/sampler.cpp, sslrec_sample_negs_mt19937): the same
        negatives bit for bit and the same generator state afterwards -- whatever draws from numpy next cannot tell the difference."""
        import ctypes as C
        from .. import _lib
        lib = _lib.load()
        n_user, n_item = self._coomat.shape[0], configs['data']['item_num']
        if not hasattr(self, '_trn_csr'):
            import scipy.sparse as sp
            csr = sp.csr_matrix((np.ones(len(self.rows), dtype=np.int8), (self.rows, self.cols)), shape=(n_user, max(n_item, self._coomat.shape[1])))
            csr.sum_duplicates()
            csr.sort_indices()
            self._trn_csr = (np.ascontiguousarray(csr.indptr, dtype=np.int64), np.ascontiguousarray(csr.indices, dtype=np.int32))
            self._rows_i32 = np.ascontiguousarray(self.rows, dtype=np.int32)
        kind, key, pos, has_gauss, cached = np.random.get_state()
        if kind != 'MT19937':
            raise RuntimeError("numpy's global generator is %r, not MT19937" % kind)
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        pos_c = C.c_int32(int(pos))
        draws = C.c_int64(0)
        negs = self.negs if (self.negs.dtype == np.int32 and self.negs.flags.c_contiguous) else np.empty(len(self.rows), dtype=np.int32)
        rowptr, col = self._trn_csr
        _lib.check(lib.sslrec_sample_negs_mt19937(key.ctypes.data, C.addressof(pos_c), self._rows_i32.ctypes.data, len(self.rows),
                                                  rowptr.ctypes.data, col.ctypes.data, n_user, n_item, negs.ctypes.data,
                                                  C.addressof(draws)), 'sslrec_sample_negs_mt19937')
        np.random.set_state((kind, key, pos_c.value, has_gauss, cached))
        if negs is not self.negs:
            self.negs[:] = negs
        self.last_sampler_draws = draws.value

    def _sample_negs_vectorized(self):
        """Opt-in (`train.fast_neg_sampling: true`) replacement for the per-interaction Python loop
        (2.2 us/edge upstream, ~5 s per epoch at amazon-book size): draw all negatives at once,
        test membership against the sorted (user, item) keys, redraw only the collisions.  Same
        distribution (uniform over the items a user has not interacted with) but a DIFFERENT random
        stream than the reference, hence off by default."""
        n_item = configs['data']['item_num']
        if not hasattr(self, '_sorted_keys'):
            self._sorted_keys = np.sort(self.rows.astype(np.int64) * n_item + self.cols.astype(np.int64))
        users = self.rows.astype(np.int64)
        negs = np.random.randint(n_item, size=users.size)
        todo = np.arange(users.size)
        while todo.size:
            keys = users[todo] * n_item + negs[todo]
            pos = np.searchsorted(self._sorted_keys, keys)
            pos[pos == self._sorted_keys.size] = 0
            clash = self._sorted_keys[pos] == keys
            todo = todo[clash]
            negs[todo] = np.random.randint(n_item, size=todo.size)
        self.negs[:] = negs

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        return self.rows[idx], self.cols[idx], self.negs[idx]


class PairwiseWEpochFlagTrnData(PairwiseTrnData):
    """flag = 1 on the very first sample ever drawn and on sample 0 of every `epoch_period`-th epoch"""

    def __init__(self, coomat):
        super().__init__(coomat)
        self.epoch_flag_counter = -1
        self.epoch_period = configs['model']['epoch_period']

    def __getitem__(self, idx):
        first_ever = self.epoch_flag_counter == -1
        if first_ever:
            self.epoch_flag_counter = 0
        period_start = False
        if idx == 0:
            self.epoch_flag_counter += 1
            period_start = self.epoch_flag_counter % self.epoch_period == 0
        return (*super().__getitem__(idx), int(first_ever or period_start))


class AllRankTstData(data.Dataset):
    def __init__(self, coomat, trn_mat):
        self.csrmat = (trn_mat.tocsr() != 0) * 1.0         # train interactions to mask at scoring time
        per_user = [[] for _ in range(coomat.shape[0])]
        for user, item in zip(coomat.row, coomat.col):
            per_user[user].append(item)
        self.user_pos_lists = per_user
        self.test_users = np.array(list(set(coomat.row.tolist())))

    def __len__(self):
        return len(self.test_users)

    def __getitem__(self, idx):
        user = self.test_users[idx]
        return user, self.csrmat[user].toarray().reshape(-1)


def sample_negs_device(users, sorted_keys, n_item, generator=None):
    """Opt-in (`train.device_sampler: true`, §8f rank 3) device-side version of `sample_negs`: one uniform
    draw per interaction, membership test against the sorted (user * n_item + item) keys with
    `torch.searchsorted`, collisions redrawn until none is left.  Same distribution as the reference's
    rejection loop (datasets_general_cf.py:13-20), different random stream.  Works on any torch device."""
    import torch
    n = users.numel()
    negs = torch.randint(n_item, (n,), device=users.device, generator=generator)
    todo = torch.arange(n, device=users.device)
    while todo.numel():
        keys = users[todo] * n_item + negs[todo]
        pos = torch.searchsorted(sorted_keys, keys).clamp_(max=sorted_keys.numel() - 1)
        todo = todo[sorted_keys[pos] == keys]
        negs[todo] = torch.randint(n_item, (todo.numel(),), device=users.device, generator=generator)
    return negs


class ExactPairwiseLoader:
    """DEFAULT train loader: `DataLoader(PairwiseTrnData, batch_size, shuffle=True, num_workers=0)`
    (data_utils/data_handler_general_cf.py:95) with the SAME batches and the SAME consumption of torch's CPU generator, without
    the 4096 `__getitem__` calls + collate per batch (4 ms of host time per 0.5 ms GPU step at amazon-book size).

    What `iter(DataLoader)` draws, in order (torch/utils/data/dataloader.py `_BaseDataLoaderIter.__init__`, sampler.py
    `RandomSampler.__iter__`): (1) at `iter()`: the loader's base seed, one int64 `random_()` from the global CPU generator;
    (2) at the first `next()`: the sampler's seed, another int64 `random_()` from the global generator, then
    `torch.randperm(n, generator=Generator().manual_seed(seed))`.  Both are issued here at the same two moments by the same
    torch calls; a batch is then three slices of the permuted arrays, as int32 tensors like `default_collate` makes of the
    dataset's numpy int32 scalars.  Same iteration protocol as a DataLoader for what the trainer touches (`len()`,
    `.dataset`, `.batch_size`; last batch short, `drop_last=False`)."""

    def __init__(self, dataset, batch_size, device=None):
        """device: when given, the epoch's permuted triples are moved there ONCE as int64 and the batches are device slices
        (the trainer's `.long().to(device)` is then a no-op): one H2D copy per epoch instead of three per step"""
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), device

    def __len__(self):
        return -(-len(self.dataset) // self.batch_size)

    def __iter__(self):
        import torch
        self._base_seed = torch.empty((), dtype=torch.int64).random_().item()      # (1), drawn when the iterator is created
        return self._batches()

    def _batches(self):
        import torch
        ds = self.dataset
        seed = int(torch.empty((), dtype=torch.int64).random_().item())             # (2), drawn when the first batch is asked for
        gen = torch.Generator()
        gen.manual_seed(seed)
        order = torch.randperm(len(ds), generator=gen)
        if self.device is not None:
            # the permutation is applied on the device: the interactions live there (int64, moved once), an epoch ships its negatives
            # and the permutation (28 MB at amazon-book size) and gathers three arrays -- the host does nothing per interaction
            if getattr(self, '_dev_src', None) is None or self._dev_src[0] is not ds.rows or self._dev_src[1] is not ds.cols:
                self._dev_src = (ds.rows, ds.cols, torch.from_numpy(np.ascontiguousarray(ds.rows)).to(self.device).long(),
                                 torch.from_numpy(np.ascontiguousarray(ds.cols)).to(self.device).long())
            order_d = order.to(self.device)
            negs_d = torch.from_numpy(np.ascontiguousarray(ds.negs)).to(self.device).long()
            cols = [self._dev_src[2][order_d], self._dev_src[3][order_d], negs_d[order_d]]
        else:
            cols = [torch.from_numpy(np.ascontiguousarray(a))[order] for a in (ds.rows, ds.cols, ds.negs)]
        for lo in range(0, len(ds), self.batch_size):
            hi = lo + self.batch_size
            yield [c[lo:hi] for c in cols]


class FastPairwiseLoader:
    """Opt-in (`train.fast_loader: true`) stand-in for `DataLoader(PairwiseTrnData, shuffle=True)`:
    a batch is three slices of the shuffled (anchor, positive, negative) arrays instead of 4096
    `__getitem__` calls + a collate (measured 4 ms of host time per batch against a 1 ms GPU step at
    amazon-book size).  Same iteration protocol (`len()`, `.dataset`, yields int32 tensors, last batch
    short); the shuffle comes from the torch CPU generator but is NOT the permutation the reference's
    DataLoader would draw, hence off by default."""

    def __init__(self, dataset, batch_size, device=None):
        """device: when given, the epoch's shuffled triples are moved there ONCE (int64, 24 B per interaction)
        and the batches are device slices -- no per-step host-to-device copies"""
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), device

    def __len__(self):
        return -(-len(self.dataset) // self.batch_size)

    def _iter_device(self):
        """shuffle and negative sampling on the device: per epoch one randperm, one sampler pass, no host work"""
        import torch
        ds, dev = self.dataset, self.device
        n_item = configs['data']['item_num']
        on_gpu = str(dev).startswith('cuda')
        if not hasattr(self, '_dev_rows'):
            self._dev_rows = torch.from_numpy(np.ascontiguousarray(ds.rows)).to(dev).long()
            self._dev_cols = torch.from_numpy(np.ascontiguousarray(ds.cols)).to(dev).long()
            if on_gpu:      # the HIP sampler (sslrec_sample_negs): Philox draws + binary search in the user's train row
                import scipy.sparse as sp
                from ..rng import PhiloxState
                csr = sp.csr_matrix((np.ones(len(ds.rows), dtype=np.int8), (ds.rows, ds.cols)),
                                    shape=(configs['data']['user_num'], n_item))
                csr.sort_indices()
                self._trn_csr = (torch.from_numpy(csr.indptr.astype(np.int64)).to(dev), torch.from_numpy(csr.indices.astype(np.int64)).to(dev))
                self._philox = PhiloxState(dev)
            else:           # same distribution with torch ops, for CPU devices (tests)
                self._dev_keys = torch.sort(self._dev_rows * n_item + self._dev_cols).values
        if on_gpu:
            from .. import ops
            self._philox.advance()
            negs = ops.sample_negs(self._dev_rows, self._trn_csr, n_item, self._philox)
        else:
            negs = sample_negs_device(self._dev_rows, self._dev_keys, n_item)
        order = torch.randperm(len(ds), device=dev)
        rows, cols, negs = self._dev_rows[order], self._dev_cols[order], negs[order]
        for lo in range(0, len(ds), self.batch_size):
            hi = lo + self.batch_size
            yield [rows[lo:hi], cols[lo:hi], negs[lo:hi]]

    def __iter__(self):
        import torch
        ds = self.dataset
        if self.device is not None and getattr(ds, 'negs_on_device', False):
            yield from self._iter_device()
            return
        order = torch.randperm(len(ds))
        rows = torch.from_numpy(np.ascontiguousarray(ds.rows))[order]
        cols = torch.from_numpy(np.ascontiguousarray(ds.cols))[order]
        negs = torch.from_numpy(ds.negs)[order]
        if self.device is not None:
            rows, cols, negs = (a.to(self.device).long() for a in (rows, cols, negs))
        for lo in range(0, len(ds), self.batch_size):
            hi = lo + self.batch_size
            yield [rows[lo:hi], cols[lo:hi], negs[lo:hi]]
