"""LightGCN on the HIP propagation path.  Same constructor, parameters (`user_embeds`,
`item_embeds`), methods and loss dictionary as the reference's models/general_cf/lightgcn.py
(:12-66); what changed is what runs underneath:

  reference                                   here
  t.concat + L x t.spmm + sum (:34-41)        one fused CSR-SpMM kernel per layer that also
                                              accumulates the layer SUM (ops.propagate_sum)
  EdgeDrop rebuilds the COO (:36-37)          keep-mask applied to the cached CSR
  3 gathers + cal_bpr_loss (:49-52)           fused gather-dot-softplus kernel
"""
import torch as t
from torch import nn

from ... import ops
from ...config.configurator import configs
from ..aug_utils import EdgeDrop
from ..base_model import BaseModel
from ..loss_utils import cal_bpr_loss_gathered, reg_params

init = nn.init.xavier_uniform_


class LightGCN(BaseModel):
    def __init__(self, data_handler):
        super().__init__(data_handler)
        self.adj = data_handler.torch_adj

        self.layer_num = configs['model']['layer_num']
        self.reg_weight = configs['model']['reg_weight']
        self.keep_rate = configs['model']['keep_rate']

        self.user_embeds = nn.Parameter(init(t.empty(self.user_num, self.embedding_size)))
        self.item_embeds = nn.Parameter(init(t.empty(self.item_num, self.embedding_size)))

        self.edge_dropper = EdgeDrop(device_rng=configs['model'].get('device_rng', False))
        self.is_training = True
        self.final_embeds = None

    def _propagate(self, adj, embeds):
        """One propagation step Y = adj @ embeds (differentiable w.r.t. embeds)."""
        return ops.spmm(adj, embeds)

    def _propagate_sum(self, adj, embeds, noises=None, eps=0.0):
        """All layers + layer sum in fused kernels: embeds + sum_l (adj^l embeds)."""
        return ops.propagate_sum(adj, embeds, self.layer_num, noises, eps)

    def forward(self, adj, keep_rate):
        if not self.is_training and self.final_embeds is not None:
            return self.final_embeds[:self.user_num], self.final_embeds[self.user_num:]
        embeds = t.concat([self.user_embeds, self.item_embeds], axis=0)
        if self.is_training:
            adj = self.edge_dropper(adj, keep_rate)
        embeds = self._propagate_sum(adj, embeds)
        self.final_embeds = embeds
        return embeds[:self.user_num], embeds[self.user_num:]

    def cal_loss(self, batch_data):
        self.is_training = True
        user_embeds, item_embeds = self.forward(self.adj, self.keep_rate)
        ancs, poss, negs = batch_data
        bpr_loss = cal_bpr_loss_gathered(user_embeds, item_embeds, ancs, poss, negs) / ancs.shape[0]
        reg_loss = self.reg_weight * reg_params(self)
        loss = bpr_loss + reg_loss
        losses = {'bpr_loss': bpr_loss, 'reg_loss': reg_loss}
        return loss, losses

    def full_predict(self, batch_data):
        user_embeds, item_embeds = self.forward(self.adj, 1.0)
        self.is_training = False
        pck_users, train_mask = batch_data
        pck_users = pck_users.long()
        pck_user_embeds = user_embeds[pck_users]
        full_preds = pck_user_embeds @ item_embeds.T
        full_preds = self._mask_predict(full_preds, train_mask)
        return full_preds
