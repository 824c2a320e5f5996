"""Shared machinery of the graph collaborative-filtering models on the HIP path: the two
embedding tables, the cached propagation plan, fused multi-layer propagation and all-rank
scoring.  (The reference repeats this per model file: lightgcn.py:12-43,58-66, simgcl.py:56-64.)"""
import torch as t
from torch import nn

from ... import ops
from ...config.configurator import configs
from ..base_model import BaseModel


class GraphCF(BaseModel):
    """user_embeds / item_embeds (xavier-uniform, users first -- the RNG order of the reference),
    `adj` = data_handler.torch_adj, `is_training` / `final_embeds` evaluation cache."""

    def __init__(self, data_handler):
        super().__init__(data_handler)
        self.adj = data_handler.torch_adj
        model_cfg = configs['model']
        self.layer_num = model_cfg['layer_num']
        self.reg_weight = model_cfg['reg_weight']
        xavier = nn.init.xavier_uniform_
        self.user_embeds = nn.Parameter(xavier(t.empty(self.user_num, self.embedding_size)))
        self.item_embeds = nn.Parameter(xavier(t.empty(self.item_num, self.embedding_size)))
        self.is_training = True
        self.final_embeds = None

    # -- propagation -------------------------------------------------------------------------
    def _propagate(self, adj, embeds):
        """one step: adj @ embeds, differentiable w.r.t. embeds (hook kept from the reference)"""
        return ops.spmm(adj, embeds)

    def _stacked_tables(self):
        return t.concat([self.user_embeds, self.item_embeds], axis=0)

    def _propagate_sum(self, adj, embeds, noises=None, eps=0.0):
        """embeds + sum_{l=1..L} P_l(adj^l embeds): one fused SpMM launch per layer"""
        return ops.propagate_sum(adj, embeds, self.layer_num, noises, eps)

    def _split(self, embeds):
        return embeds[:self.user_num], embeds[self.user_num:]

    def _cached(self):
        """the embeddings computed by the last forward, reused across evaluation batches"""
        if not self.is_training and self.final_embeds is not None:
            return self._split(self.final_embeds)
        return None

    # -- all-rank scoring --------------------------------------------------------------------
    def _score_all_items(self, user_embeds, item_embeds, batch_data):
        pck_users, train_mask = batch_data
        scores = user_embeds[pck_users.long()] @ item_embeds.T
        return self._mask_predict(scores, train_mask)
