"""Augmentation modules feeding the hot path; same class names and call signatures as the
reference's models/aug_utils.py (EdgeDrop :11-31, EmbedPerturb :118-132, SvdDecomposition
:82-98).  The random draws are taken from the global torch CPU generator in exactly the
reference's order and shapes (parity mode), unless `device_rng=True` (perf mode: drawn on the
GPU, statistically equivalent, not bit-equal).
"""
import torch as t
import torch.nn.functional as F
from torch import nn

from ..graph import DroppedView, graph_of


class EdgeDrop(nn.Module):
    """Drop edges of the adjacency.  Returns a `DroppedView` (the cached CSR plus the keep mask)
    instead of a rebuilt sparse tensor; `_propagate` / ops.spmm accept it wherever the
    reference passes the new adjacency.  `.to_torch_sparse()` materializes the reference's
    result on demand."""

    def __init__(self, resize_val=False, device_rng=False):
        super().__init__()
        self.resize_val = resize_val
        self.device_rng = device_rng

    def forward(self, adj, keep_rate):
        if keep_rate == 1.0:
            return adj
        graph = graph_of(adj)
        if self.device_rng:
            draw = t.rand(graph.nnz, device=graph.device)
        else:
            # same draw as the reference (CPU generator, aug_utils.py:28); only the draw crosses PCIe,
            # the threshold arithmetic (identical in fp32) runs on the device
            draw = t.rand(t.Size([graph.nnz])).to(graph.device)
        mask = (draw + keep_rate).floor().type(t.bool)
        return DroppedView(graph, mask, 1.0 / keep_rate if self.resize_val else 1.0)


class EmbedPerturb(nn.Module):
    """E + eps * sign(E) * normalize(U[0,1)^{N x d}, dim=1).  `draw(shape, device)` produces the
    noise the fused SpMM epilogue consumes; calling the module applies the perturbation to a
    dense tensor the reference's way (used outside the fused path)."""

    def __init__(self, eps, device_rng=False):
        super().__init__()
        self.eps = eps
        self.device_rng = device_rng

    def draw(self, shape, device):
        if self.device_rng:
            return t.rand(shape, device=device)
        return t.rand(shape).to(device)                    # CPU generator, like aug_utils.py:130

    def forward(self, embeds):
        noise = (F.normalize(self.draw(embeds.shape, embeds.device), p=2) * t.sign(embeds)) * self.eps
        return embeds + noise


class SvdDecomposition(nn.Module):
    """Rank-q SVD factors of the adjacency for LightGCL's second view (reference :82-98):
    returns (U^T, V^T, U*S, V*S)."""

    def __init__(self, svd_q):
        super().__init__()
        self.svd_q = svd_q

    def forward(self, adj):
        svd_u, s, svd_v = t.svd_lowrank(adj, q=self.svd_q)
        u_mul_s = svd_u @ t.diag(s)
        v_mul_s = svd_v @ t.diag(s)
        return svd_u.T, svd_v.T, u_mul_s, v_mul_s
