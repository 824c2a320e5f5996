"""SimGCL on the HIP path; interface of the reference's models/general_cf/simgcl.py (:11-64).
Two perturbed forwards (uniform-noise augmentation fused into the SpMM epilogue) + one clean
forward, two InfoNCE terms (:49).  Noise is drawn per layer per perturbed view, view 1 first
(:41-42, :25-27), from the CPU generator unless model.device_rng is set (then the noise rows are
computed inside the SpMM epilogue and never exist as tensors: sslrec_amd/rng.py)."""
from ...config.configurator import configs
from ..aug_utils import EmbedPerturb
from ..loss_utils import cal_bpr_loss_stacked, cal_infonce_loss_two_sided
from .lightgcn import LightGCN


class SimGCL(LightGCN):
    def __init__(self, data_handler):
        super().__init__(data_handler)
        self.cl_weight = configs['model']['cl_weight']
        self.temperature = configs['model']['temperature']
        self.eps = configs['model']['eps']
        self.embed_perturb = EmbedPerturb(eps=self.eps, device_rng=self.device_rng)

    def forward(self, adj, perturb=False):
        if not perturb:
            return super().forward(adj, 1.0)
        embeds = self._stacked_tables(alias_ok=True)
        noises = [self.embed_perturb.draw(embeds.shape, embeds.device) for _ in range(self.layer_num)]
        return self._split(self._propagate_sum(adj, embeds, noises, self.eps))

    def _three_views(self):
        """(stacked [users; items] tables) the two perturbed forwards and the clean one (reference :41-43) as ONE fused call: all three start from
        the same A.E0, so the first layer is a single SpMM with three epilogues (ops.propagate_sum_views); noise is
        drawn in the reference's order -- view 1's layers, then view 2's"""
        from ... import ops
        embeds = self._stacked_tables(alias_ok=True)
        draws = [[self.embed_perturb.draw(embeds.shape, embeds.device) for _ in range(self.layer_num)] for _ in range(2)]
        if self._hook_overridden():      # a plugin's own _propagate: three separate layer loops, like the reference
            return tuple(self._propagate_sum(self.adj, embeds, nz, self.eps) for nz in (draws[0], draws[1], None))
        return tuple(ops.propagate_sum_views(self.adj, embeds, self.layer_num, [draws[0], draws[1], None], self.eps))

    def cal_loss(self, batch_data):
        self.is_training = True
        self._begin_step()
        ancs, poss, negs = batch_data
        from ... import ops
        if (self.user_embeds.is_cuda and not self._hook_overridden() and len(list(self.parameters())) == 2 and
                ops.contrastive_step_ok((ops._as_adj(self.adj),), self.embedding_size, self.user_num + self.item_num, (ancs.numel(), poss.numel()))):
            # the whole step as ONE autograd node with a hand-written backward (ops.contrastive_step): same kernels, no stock launch between them
            shape = (self.user_num + self.item_num, self.embedding_size)
            draws = [[self.embed_perturb.draw(shape, self.user_embeds.device) for _ in range(self.layer_num)] for _ in range(2)]      # reference order (:41-42)
            loss, bpr_loss, cl_loss, reg_loss = ops.contrastive_step(self.user_embeds, self.item_embeds, dict(
                kind='simgcl', adj=ops._as_adj(self.adj), noises=draws, eps=self.eps, layer_num=self.layer_num, ancs=ancs, poss=poss, negs=negs,
                items_cl=poss, temp=self.temperature, cl_weight=self.cl_weight, reg_weight=self.reg_weight, precision=self.infonce_precision))
            return loss, {'bpr_loss': bpr_loss, 'reg_loss': reg_loss, 'cl_loss': cl_loss}
        view1, view2, view3 = self._three_views()      # stacked tables: the losses address user / item rows through offsets

        bpr_loss = cal_bpr_loss_stacked(view3, self.user_num, ancs, poss, negs, divisor=ancs.shape[0])
        cl_loss = cal_infonce_loss_two_sided(view1, view2, self.user_num, ancs, poss, self.temperature, self.infonce_precision)
        cl_loss = cl_loss / ancs.shape[0]
        reg_loss = self._table_regularizer()
        cl_loss = cl_loss * self.cl_weight
        loss = bpr_loss + reg_loss + cl_loss
        losses = {'bpr_loss': bpr_loss, 'reg_loss': reg_loss, 'cl_loss': cl_loss}
        return loss, losses

    def _embeddings_for_eval(self):
        self._stacked_e0 = None          # (evaluation never reuses a training step's concatenated tables)
        tables = self.forward(self.adj, False)
        self.is_training = False
        return tables

    def full_predict(self, batch_data):
        users, items = self._embeddings_for_eval()
        return self._score_all_items(users, items, batch_data)
