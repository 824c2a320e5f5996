"""LightGCN on the HIP propagation path.  Constructor argument, parameter names, `forward(adj,
keep_rate)`, `cal_loss(batch)`, `full_predict(batch)` and the loss dictionary follow the
reference's models/general_cf/lightgcn.py (:12-66); underneath,

  reference                                   here
  t.concat + L x t.spmm + sum (:34-41)        one fused CSR-SpMM launch per layer that also
                                              accumulates the layer SUM (ops.propagate_sum)
  EdgeDrop rebuilds the COO (:36-37)          keep-mask packed into the cached streamed CSR
  3 gathers + cal_bpr_loss (:49-52)           fused gather-dot-softplus kernel
"""
from ...config.configurator import configs
from ..aug_utils import EdgeDrop
from ..loss_utils import cal_bpr_loss_stacked, reg_params
from ._graph_cf import GraphCF


class LightGCN(GraphCF):
    def __init__(self, data_handler):
        super().__init__(data_handler)
        self.keep_rate = configs['model']['keep_rate']
        self.edge_dropper = EdgeDrop(device_rng=self.device_rng)

    def forward(self, adj, keep_rate, with_reg=False):
        cached = self._cached()
        if cached is not None:
            return cached
        if self.is_training:                       # LightGCN itself trains with edge dropout (lightgcn.yml)
            adj = self.edge_dropper(adj, keep_rate)
        if with_reg:      # cal_loss: the regularizer of the two tables (their only parameters) on the propagation's autograd node
            self.final_embeds, self._reg_loss = self._propagate_sum(adj, self._stacked_tables(alias_ok=True), reg_weight=self.reg_weight)
        else:
            self.final_embeds = self._propagate_sum(adj, self._stacked_tables(alias_ok=True))
        return self._split(self.final_embeds)

    def cal_loss(self, batch_data):
        self.is_training = True
        self._begin_step()
        fused_reg = len(list(self.parameters())) == 2      # a subclass with further parameters: reg_params over all of them
        self.forward(self.adj, self.keep_rate, with_reg=fused_reg)
        ancs, poss, negs = batch_data
        reg_loss = self._reg_loss if fused_reg else reg_params(self, self.reg_weight)
        # `bpr_loss + reg_loss` (reference :54) comes out of the BPR kernel's own finishing step: no elementwise launch for the sum
        loss, bpr_loss = cal_bpr_loss_stacked(self.final_embeds, self.user_num, ancs, poss, negs, divisor=ancs.shape[0], add=reg_loss)
        return loss, {'bpr_loss': bpr_loss, 'reg_loss': reg_loss}

    def _embeddings_for_eval(self):
        self._stacked_e0 = None          # (evaluation never reuses a training step's concatenated tables)
        tables = self.forward(self.adj, 1.0)
        self.is_training = False
        return tables

    def full_predict(self, batch_data):
        users, items = self._embeddings_for_eval()
        return self._score_all_items(users, items, batch_data)
