"""Augmentation modules feeding the hot path; same class names and call signatures as the
reference's models/aug_utils.py (EdgeDrop :11-31, EmbedPerturb :118-132, SvdDecomposition
:82-98).  The random draws are taken from the global torch CPU generator in exactly the
reference's order and shapes (parity mode), unless `device_rng` is given (perf mode: a
`sslrec_amd.rng.PhiloxState`; the kernels COMPUTE mask bits / noise rows in place, nothing is drawn, stored or
copied -- statistically equivalent, not bit-equal).  With `sslrec_amd.rng.enable_host_replay` (the Trainer's default on a
GPU) the parity-mode draws are the same numbers, but produced by the CPU generator's algorithm running on the device.
"""
import torch as t
import torch.nn.functional as F
from torch import nn

from ..graph import DroppedView, graph_of


class _PinnedDraws:
    """Parity mode keeps the reference's CPU draws (`t.rand` on the global CPU generator, same shapes, same order:
    aug_utils.py:28,130) but not its blocking pageable copy: the numbers are drawn straight into page-locked staging
    buffers (a small ring per shape; a buffer is reused only after its last copy has finished) and copied with an
    asynchronous DMA, so the next draw on the host overlaps the previous draw's transfer and the kernels already
    enqueued.  The values and the generator state are exactly those of `t.rand(shape)`."""

    def __init__(self, depth=3):
        self.depth, self.rings = depth, {}

    def rand_to(self, shape, device):
        device = t.device(device)
        shape = tuple(shape)
        if device.type != 'cuda':
            return t.rand(shape).to(device)
        from ..rng import active_host_replay
        replay = active_host_replay(device)
        if replay is not None:       # the same numbers, generated on the device (sslrec_amd/csrc/mt19937.hip)
            return replay.rand(shape)
        ring = self.rings.setdefault((shape, device.index), {'bufs': [], 'events': [], 'next': 0})
        i = ring['next'] % self.depth
        ring['next'] += 1
        if i >= len(ring['bufs']):
            ring['bufs'].append(t.empty(shape, dtype=t.float32).pin_memory())
            ring['events'].append(None)
        elif ring['events'][i] is not None:
            ring['events'][i].synchronize()
        drawn = t.rand(shape, out=ring['bufs'][i])
        if drawn is not ring['bufs'][i]:                  # a stand-in generator (tests replay recorded draws)
            ring['bufs'][i].copy_(drawn)
        out = ring['bufs'][i].to(device, non_blocking=True)
        ev = t.cuda.Event()
        ev.record()
        ring['events'][i] = ev
        return out


_pinned = _PinnedDraws()


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
        scale = 1.0 / keep_rate if self.resize_val else 1.0
        if self.device_rng:      # the compaction kernels compute floor(u_k + keep_rate) per COO entry k (Philox)
            return DroppedView(graph, None, scale, philox=(self.device_rng, self.device_rng.next_stream(), keep_rate))
        # same draw as the reference (CPU generator, aug_utils.py:28); only the draw crosses PCIe,
        # the threshold arithmetic (identical in fp32) runs on the device
        from ..rng import active_host_replay
        replay = active_host_replay(graph.device)
        if replay is not None:       # the generator itself runs on the device: the mask is written directly
            return DroppedView(graph, replay.keep_mask(graph.nnz, keep_rate), scale)
        draw = _pinned.rand_to((graph.nnz,), graph.device)
        mask = (draw + keep_rate).floor().type(t.bool)
        return DroppedView(graph, mask, scale)


class EmbedPerturb(nn.Module):
    """E + eps * sign(E) * normalize(U[0,1)^{N x d}, dim=1).  `draw(shape, device)` produces the
    noise the fused SpMM epilogue consumes; calling the module applies the perturbation to a
    dense tensor the reference's way (used outside the fused path)."""

    def __init__(self, eps, device_rng=False):
        super().__init__()
        self.eps = eps
        self.device_rng = device_rng

    def draw(self, shape, device):
        if self.device_rng:      # a token: the SpMM epilogue computes the rows (sslrec_amd.rng.PhiloxNoise)
            from ..rng import PhiloxNoise
            return PhiloxNoise(self.device_rng, shape)
        return _pinned.rand_to(shape, device)              # CPU generator, like aug_utils.py:130

    def forward(self, embeds):
        u = self.draw(embeds.shape, embeds.device)
        if not t.is_tensor(u):
            u = u.materialize()
        noise = (F.normalize(u, p=2) * t.sign(embeds)) * self.eps
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
