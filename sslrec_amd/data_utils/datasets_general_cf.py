"""Training / evaluation datasets of the general-CF scenario.  Same classes, attributes and
RNG consumption as the reference's data_utils/datasets_general_cf.py:6-68 so that a fixed
seed yields the same batches:

  * PairwiseTrnData      -- BPR triples; `sample_negs()` draws one negative per interaction
                            from numpy's GLOBAL generator by rejection, in interaction order
                            (reference :13-20);
  * AllRankTstData       -- one (user, dense train-mask row) pair per test user (reference :46-68).
"""
import numpy as np
import torch.utils.data as data

from ..config.configurator import configs


class PairwiseTrnData(data.Dataset):
    def __init__(self, coomat):
        self.rows = coomat.row
        self.cols = coomat.col
        self.dokmat = coomat.todok()
        self.negs = np.zeros(len(self.rows)).astype(np.int32)

    def sample_negs(self):
        item_num = configs['data']['item_num']
        seen = self.dokmat
        for i, u in enumerate(self.rows):
            neg = np.random.randint(item_num)
            while (u, neg) in seen:
                neg = np.random.randint(item_num)
            self.negs[i] = neg

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        return self.rows[idx], self.cols[idx], self.negs[idx]


class PairwiseWEpochFlagTrnData(PairwiseTrnData):
    """NCL's variant (reference :28-44): additionally yields a flag that is 1 on the very first
    sample and whenever a new epoch whose index is a multiple of `epoch_period` starts."""

    def __init__(self, coomat):
        super().__init__(coomat)
        self.epoch_flag_counter = -1
        self.epoch_period = configs['model']['epoch_period']

    def __getitem__(self, idx):
        flag = 0
        if self.epoch_flag_counter == -1:
            flag, self.epoch_flag_counter = 1, 0
        if idx == 0:
            self.epoch_flag_counter += 1
            if self.epoch_flag_counter % self.epoch_period == 0:
                flag = 1
        anc, pos, neg = super().__getitem__(idx)
        return anc, pos, neg, flag


class AllRankTstData(data.Dataset):
    def __init__(self, coomat, trn_mat):
        self.csrmat = (trn_mat.tocsr() != 0) * 1.0
        user_pos_lists = [list() for _ in range(coomat.shape[0])]
        test_users = set()
        for row, col in zip(coomat.row, coomat.col):
            user_pos_lists[row].append(col)
            test_users.add(row)
        self.test_users = np.array(list(test_users))
        self.user_pos_lists = user_pos_lists

    def __len__(self):
        return len(self.test_users)

    def __getitem__(self, idx):
        user = self.test_users[idx]
        mask = np.reshape(self.csrmat[user].toarray(), [-1])
        return user, mask
