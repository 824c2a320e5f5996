"""LightGCL on the HIP path; interface of the reference's models/general_cf/lightgcl.py
(:12-144): parameters user_embeds / item_embeds / Ws.{l}.W, `forward(test=False)`,
`cal_loss`, `full_predict`.

  reference                                          here
  per-nnz Python normalization loop (:17-20)         vectorized 1/sqrt(d_u d_i) in fp32 (the same bits as
                                                     the scalar powf loop on the tiny and the yelp data)
  `_spmm`: coalesce + gather nnz x d + index_add_    the CSR SpMM kernel on a CSR of A (U x I)
  with atomics, twice per layer (:58-65, :78-79)     and one of A^T, both built once
  B x U and B x I score matrices (:114-117)          fused un-normalized InfoNCE (variant 1)
  `u_mul_s @ (vt @ E)` as two skinny GEMMs (:82-85)   two rank-q streaming kernels (ops.lowrank_apply)
The one-time `svd_lowrank` (:25) stays on PyTorch (K10 of SURVEY.md §2.3, not on the roofline of this path).
"""
import numpy as np
import torch as t
from torch import nn

from ... import ops
from ...config.configurator import configs
from ...graph import PropGraph, RevaluedView
from ..aug_utils import SvdDecomposition
from ..base_model import BaseModel
from ._graph_cf import GraphCF
from ..loss_utils import reg_params

init = nn.init.xavier_uniform_


class LightGCL(BaseModel):
    def __init__(self, data_handler):
        super().__init__(data_handler)
        device = configs['device']
        train_mat = data_handler._load_one_mat(data_handler.trn_file)        # binarized float32 COO
        rows, cols = train_mat.row.astype(np.int64), train_mat.col.astype(np.int64)
        row_d = np.asarray(train_mat.sum(1)).reshape(-1).astype(np.float32)
        col_d = np.asarray(train_mat.sum(0)).reshape(-1).astype(np.float32)
        vals = (train_mat.data.astype(np.float32) / np.sqrt(row_d[rows] * col_d[cols])).astype(np.float32)
        # the reference coalesces (sorts by (row, col)); keep that entry order for the dropout draws
        order = np.lexsort((cols, rows))
        rows, cols, vals = rows[order], cols[order], vals[order]
        idx = t.from_numpy(np.vstack((rows, cols)))
        self.adj = t.sparse_coo_tensor(idx, t.from_numpy(vals), train_mat.shape, is_coalesced=True).to(device)
        self.graph = PropGraph(rows, cols, vals, train_mat.shape, device)    # CSR of A and of A^T

        self.svd_decompose = SvdDecomposition(svd_q=configs['model']['svd_q'])
        self.ut, self.vt, self.u_mul_s, self.v_mul_s = (x.contiguous() for x in self.svd_decompose(self.adj))

        self.temp = configs['model']['temp']
        self.dropout = configs['model']['dropout']
        self.layer_num = configs['model']['layer_num']
        self.cl_weight = configs['model']['cl_weight']
        self.reg_weight = configs['model']['reg_weight']
        self.infonce_precision = configs['model'].get('infonce_precision') or None      # see GraphCF.__init__

        self.user_embeds = nn.Parameter(init(t.empty(self.user_num, self.embedding_size)))
        self.item_embeds = nn.Parameter(init(t.empty(self.item_num, self.embedding_size)))
        self.E_u = None
        self.E_i = None
        self.G_u = None
        self.G_i = None
        self.act = nn.LeakyReLU(0.5)
        self.Ws = nn.ModuleList([W_contrastive(self.embedding_size) for _ in range(self.layer_num)])
        self.is_training = True

    def _spmm(self, sp, emb):
        """sp @ emb for `sp` = the U x I graph view or its transpose."""
        return ops.spmm(sp, emb)

    @staticmethod
    def _lowrank(left, right, emb):
        """left @ (right @ emb) through the two rank-q streaming kernels (model.svd_q <= 16)"""
        return ops.lowrank_apply(left, right, emb)

    def _sparse_dropout(self, graph, dropout):
        """Dropout on the adjacency VALUES (reference :67-71; applied in training mode always,
        as upstream calls F.dropout with its default training=True)."""
        if dropout == 0:
            return graph
        return RevaluedView(graph, nn.functional.dropout(self.adj.values(), p=dropout))

    def forward(self, test=False):
        if test and self.E_u is not None:
            return self.E_u, self.E_i
        e_u, e_i = [self.user_embeds], [self.item_embeds]
        g_u, g_i = [self.user_embeds], [self.item_embeds]
        for _ in range(self.layer_num):
            z_u = self._spmm(self._sparse_dropout(self.graph, self.dropout), e_i[-1])
            z_i = self._spmm(self._sparse_dropout(self.graph, self.dropout).transposed(), e_u[-1])
            g_u.append(self._lowrank(self.u_mul_s, self.vt, e_i[-1]))       # u_mul_s @ (vt @ E_i), reference :83
            g_i.append(self._lowrank(self.v_mul_s, self.ut, e_u[-1]))
            e_u.append(z_u)
            e_i.append(z_i)
        self.G_u, self.G_i = sum(g_u), sum(g_i)
        self.E_u, self.E_i = sum(e_u), sum(e_i)
        return self.E_u, self.E_i

    def cal_loss(self, batch_data):
        self.is_training = True
        user_embeds, item_embeds = self.forward()
        ancs, poss, negs = batch_data
        bsz = ancs.shape[0]
        bpr_loss = ops.bpr_loss_gathered(user_embeds, item_embeds, ancs, poss, negs, variant=1, divisor=bsz)
        cl_loss = (ops.infonce_loss_gathered(self.G_u, self.E_u, ancs, self.temp, variant=1, precision=self.infonce_precision) +
                   ops.infonce_loss_gathered(self.G_i, self.E_i, poss, self.temp, variant=1, precision=self.infonce_precision)) / bsz
        reg_loss = reg_params(self, self.reg_weight)
        cl_loss = self.cl_weight * cl_loss
        loss = bpr_loss + cl_loss + reg_loss
        losses = {'bpr_loss': bpr_loss, 'reg_loss': reg_loss, 'cl_loss': cl_loss}
        return loss, losses

    # device-side evaluation (trainer/metrics.py): the same CSR-masked top-k as the other graph models
    def _embeddings_for_eval(self):
        tables = self.forward(test=True)
        self.is_training = False
        return tables

    predict_topk = GraphCF.predict_topk
    _score_all_items = GraphCF._score_all_items

    def full_predict(self, batch_data):
        user_embeds, item_embeds = self._embeddings_for_eval()
        return self._score_all_items(user_embeds, item_embeds, batch_data)


class W_contrastive(nn.Module):
    """d x d matrix that the upstream forward never applies but `reg_params` regularizes and
    the state_dict carries (reference :138-144)."""

    def __init__(self, d):
        super().__init__()
        self.W = nn.Parameter(nn.init.xavier_uniform_(t.empty(d, d)))

    def forward(self, x):
        return x @ self.W
