"""SGL (edge-drop augmentation) on the HIP path; interface of the reference's
models/general_cf/sgl.py (:11-65).  Two independently edge-dropped views + the clean view,
three InfoNCE terms against ALL users / items of view 2 (:57-59), each a fused
gather-normalize-MFMA-logsumexp kernel.  Only `augmentation: edge_drop` (the configured
default, sgl.yml) is supported -- `random_walk` raises and `node_drop` mixes devices in the
reference itself."""
from ...config.configurator import configs
from ..loss_utils import cal_bpr_loss_stacked, cal_infonce_loss_two_sided
from .lightgcn import LightGCN


class SGL(LightGCN):
    def __init__(self, data_handler):
        super().__init__(data_handler)
        self.augmentation = configs['model']['augmentation']
        self.cl_weight = configs['model']['cl_weight']
        self.temperature = configs['model']['temperature']
        if self.augmentation != 'edge_drop':
            raise NotImplementedError("SGL augmentation '%s': only 'edge_drop' is functional" % self.augmentation)

    def forward(self, adj, keep_rate):
        cached = self._cached()
        if cached is not None:
            return cached
        adj = self.edge_dropper(adj, keep_rate)          # one mask per view, shared by all layers (:27-28)
        self.final_embeds = self._propagate_sum(adj, self._stacked_tables(alias_ok=True))
        return self._split(self.final_embeds)

    def cal_loss(self, batch_data):
        self.is_training = True
        self._begin_step()
        keep_rate = configs['model']['keep_rate']
        ancs, poss, negs = batch_data
        import torch as t
        from ... import ops
        if self.user_embeds.is_cuda and not self._hook_overridden() and len(list(self.parameters())) == 2:
            adj1 = self.edge_dropper(self.adj, keep_rate)      # one mask per view, in the reference's order (:47-48)
            adj2 = self.edge_dropper(self.adj, keep_rate)
            adjs = tuple(ops._as_adj(a) for a in (adj1, adj2, self.adj))
            if ops.contrastive_step_ok(adjs, self.embedding_size, self.user_num + self.item_num, (ancs.numel(), 2 * poss.numel())):
                # the whole step as ONE autograd node with a hand-written backward (ops.contrastive_step)
                loss, bpr_loss, cl_loss, reg_loss = ops.contrastive_step(self.user_embeds, self.item_embeds, dict(
                    kind='sgl', adjs=adjs, layer_num=self.layer_num, ancs=ancs, poss=poss, negs=negs, items_cl=t.cat([poss, negs]),
                    temp=self.temperature, cl_weight=self.cl_weight, reg_weight=self.reg_weight, precision=self.infonce_precision))
                return loss, {'bpr_loss': bpr_loss, 'reg_loss': reg_loss, 'cl_loss': cl_loss}
            return self._cal_loss_separate_nodes(batch_data, (adj1, adj2))
        return self._cal_loss_separate_nodes(batch_data, None)

    def _cal_loss_separate_nodes(self, batch_data, dropped):
        """the step composed from the separate autograd nodes (rounds 1-4; also what a plugin's own _propagate gets)"""
        keep_rate = configs['model']['keep_rate']
        # the three views as STACKED [users; items] tables (what the propagation returns): the losses address the user / item rows
        # through offsets, so no slice of a table enters the autograd graph (each would cost a table-sized zero fill + copy backward)
        if dropped is None:
            self.forward(self.adj, keep_rate)
            view1 = self.final_embeds
            self.forward(self.adj, keep_rate)
            view2 = self.final_embeds
        else:           # (the masks were already drawn, in the reference's order)
            view1 = self._propagate_sum(dropped[0], self._stacked_tables(alias_ok=True))
            view2 = self._propagate_sum(dropped[1], self._stacked_tables(alias_ok=True))
        self.forward(self.adj, 1.0)
        view3 = self.final_embeds
        ancs, poss, negs = batch_data

        bpr_loss = cal_bpr_loss_stacked(view3, self.user_num, ancs, poss, negs, divisor=ancs.shape[0])
        # the two item-side terms (:58-59) score their anchors against the SAME `all` operand (view 2's item table) and the loss
        # is a plain sum over anchors: one call over the 2B anchors [poss; negs] prepares / splits / streams that table once
        import torch as t
        cl_loss = cal_infonce_loss_two_sided(view1, view2, self.user_num, ancs, t.cat([poss, negs]), self.temperature, self.infonce_precision)
        cl_loss = cl_loss / ancs.shape[0]
        reg_loss = self._table_regularizer()
        cl_loss = cl_loss * self.cl_weight
        loss = bpr_loss + reg_loss + cl_loss
        losses = {'bpr_loss': bpr_loss, 'reg_loss': reg_loss, 'cl_loss': cl_loss}
        return loss, losses
