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
        self.dokmat = coomat.todok()                       # O(1) membership for the rejection test
        self.negs = np.zeros(len(self.rows)).astype(np.int32)

    def sample_negs(self):
        n_item = configs['data']['item_num']
        interacted = self.dokmat
        draw = np.random.randint
        for pos, user in enumerate(self.rows):
            candidate = draw(n_item)
            while (user, candidate) in interacted:
                candidate = draw(n_item)
            self.negs[pos] = candidate

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
