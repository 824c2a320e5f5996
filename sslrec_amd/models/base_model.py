"""BaseModel: the plugin contract of the reference's models/base_model.py:6-46
(forward / cal_loss(batch) -> (loss, dict) / full_predict((users, train_mask)) -> [B, I],
`_mask_predict`), unchanged."""
from torch import nn

from ..config.configurator import configs


class BaseModel(nn.Module):
    def __init__(self, data_handler):
        super().__init__()
        self.user_num = configs['data']['user_num']
        self.item_num = configs['data']['item_num']
        self.embedding_size = configs['model']['embedding_size']

    def forward(self):
        pass

    def cal_loss(self, batch_data):
        """-> (0-d loss tensor, {name: loss term})"""
        pass

    def _mask_predict(self, full_preds, train_mask):
        return full_preds * (1 - train_mask) - 1e8 * train_mask

    def full_predict(self, batch_data):
        """-> [test_batch, item_num] scores with training items pushed to -1e8"""
        pass
