"""BaseModel -- the plugin contract every recommender implements (reference models/base_model.py:6-46):

    forward(...)                      model-specific, returns embeddings
    cal_loss(batch) -> (loss, parts)  0-d differentiable loss + {name: term} for logging
    full_predict((users, train_mask)) -> [len(users), item_num] scores, training items at -1e8
    _mask_predict(scores, train_mask) the masking rule shared by all models

Sizes come from the global configuration that the data handler completes at load time."""
from torch import nn

from ..config.configurator import configs


class BaseModel(nn.Module):
    def __init__(self, data_handler):
        super().__init__()
        data_cfg = configs['data']
        self.user_num, self.item_num = data_cfg['user_num'], data_cfg['item_num']
        self.embedding_size = configs['model']['embedding_size']

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def cal_loss(self, batch_data):
        raise NotImplementedError

    def full_predict(self, batch_data):
        raise NotImplementedError

    def _mask_predict(self, full_preds, train_mask):
        """keep scores of unseen items, push items seen in training to -1e8"""
        return full_preds * (1 - train_mask) - 1e8 * train_mask
