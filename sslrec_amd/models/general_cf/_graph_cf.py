"""Shared machinery of the graph collaborative-filtering models on the HIP path: the two
embedding tables, the cached propagation plan, fused multi-layer propagation and all-rank
scoring.  (The reference repeats this per model file: lightgcn.py:12-43,58-66, simgcl.py:56-64.)"""
import torch as t
from torch import nn

from ... import ops
from ...config.configurator import configs
from ..base_model import BaseModel

EVAL_DENSE_CHUNK = 1024      # users per dense score matrix when predict_topk cannot take the fused kernel (the reference's test batch size)


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
        # The two parameters are adjacent row ranges of ONE buffer -- [user_embeds; item_embeds], the table the propagation works on
        # (reference lightgcn.py:34 concatenates them in every forward) -- initialised in the reference's order with the reference's
        # draws (a row range of a contiguous table is itself contiguous: xavier_uniform_ sees the same shape and draws the same numbers).
        # `_stacked_tables` then needs no copy; `_apply` (.to / .cuda) and anything else that re-seats a parameter's storage is detected.
        table = t.empty(self.user_num + self.item_num, self.embedding_size)
        self.user_embeds = nn.Parameter(xavier(table[:self.user_num]))
        self.item_embeds = nn.Parameter(xavier(table[self.user_num:]))
        self.is_training = True
        self.final_embeds = None
        # opt-in perf switch like model.device_rng: arithmetic of the fused InfoNCE products ('x6' default with
        # fp32-level error, 'fp32' exact, 'x36' / 'x3' faster with reduced score precision; csrc/infonce_x3.inc).
        # Handed to every InfoNCE call of THIS model (forward and backward use the same mode); no process-wide state.
        self.infonce_precision = model_cfg.get('infonce_precision') or None
        # opt-in perf switch: augmentation randomness computed in the kernels (sslrec_amd/rng.py) instead of the
        # reference's CPU draws; None = parity mode
        self.device_rng = None
        if model_cfg.get('device_rng'):
            from ...rng import PhiloxState
            self.device_rng = PhiloxState(configs['device'])

    def _begin_step(self):
        """start of a training forward: a fresh RNG step for the device-side augmentations (capturable kernel)"""
        # The evaluation cache of the PREVIOUS step still references that step's autograd graph, and with it the parameters'
        # AccumulateGrad nodes -- which remember the stream they were created on and would be reused by this forward.  Steps run on
        # the default stream and then captured into a hipGraph on another one would run AccumulateGrad on the legacy default stream
        # inside the capture (hipStreamEndCapture then crashes; tools/capture_probe.py).  Dropping the reference first lets the nodes
        # die with their graph.
        self.final_embeds = None
        self._stacked_e0 = None          # the concatenation of the two parameter tables, made once per training step
        if hasattr(self, '_reg_loss'):
            self._reg_loss = None
        if self.device_rng is not None:
            self.device_rng.advance()
        else:       # parity mode with the generator replayed on the device: a step boundary for its draw-ahead
            from ...rng import active_host_replay
            rep = active_host_replay(self.user_embeds.device)
            if rep is not None:
                rep.begin_step()

    # -- propagation -------------------------------------------------------------------------
    def _propagate(self, adj, embeds):
        """one step: adj @ embeds, differentiable w.r.t. embeds (the hook of reference lightgcn.py:28-29).  The in-tree
        models run the fused multi-layer form below; a subclass that OVERRIDES this hook is honoured: `_propagate_sum`
        then falls back to the reference's layer loop around it."""
        return ops.spmm(adj, embeds)

    def _hook_overridden(self):
        return type(self)._propagate is not GraphCF._propagate

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._restack()          # (.to / .cuda / .float give every parameter a storage of its own)
        return out

    def _restack(self):
        u, i = self.user_embeds, self.item_embeds
        if ops.stacked_alias(u, i) is None and u.dtype == i.dtype and u.device == i.device and u.dim() == 2 and u.shape[1] == i.shape[1]:
            table = t.cat([u.data, i.data])
            u.data, i.data = table[:u.shape[0]], table[u.shape[0]:]

    def _stacked_tables(self, alias_ok=False):
        """[user_embeds; item_embeds] (reference lightgcn.py:34: a fresh `t.concat`, i.e. a COPY the caller may do anything with).
        alias_ok=True -- passed only by the in-tree fused paths, which never write to the table -- returns an ALIAS of the buffer the two
        parameters share (see __init__), joined to autograd by ops.stack_params (no copy forward, row ranges of the gradient backward);
        inside a training step (between `_begin_step` calls) that tensor is made once: SGL's three views and the stacked regularizer
        share it.  Everybody else -- a subclass's own forward, a plugin `_propagate` hook (`_hook_overridden`) -- gets the reference's
        copy: an in-place op on it (masking, normalisation) cannot reach the parameters."""
        if not alias_ok or self._hook_overridden():
            return t.concat([self.user_embeds, self.item_embeds], axis=0)
        if not self.is_training or not t.is_grad_enabled():
            alias = ops.stacked_alias(self.user_embeds, self.item_embeds)
            return alias if alias is not None else t.concat([self.user_embeds, self.item_embeds], axis=0)
        stamp = (self.user_embeds._version, self.item_embeds._version, self.user_embeds.data_ptr(), self.item_embeds.data_ptr())
        cached = getattr(self, '_stacked_e0', None)
        if cached is None or cached[0] != stamp:      # (an optimizer step or any other in-place write bumps the version counters)
            cached = self._stacked_e0 = (stamp, ops.stack_params(self.user_embeds, self.item_embeds))
        return cached[1]

    def _table_regularizer(self):
        """reg_params(self) * reg_weight (loss_utils.py:20-24, lightgcn.py:53).  The two embedding tables being the model's only
        parameters, it is ONE sum of squares over their concatenation (one launch forward, one backward, one gradient to add to the
        propagation's) instead of a launch pair per parameter; any further parameter: reg_params over all of them."""
        from ..loss_utils import reg_params
        if len(list(self.parameters())) == 2 and self.user_embeds.is_cuda:
            return ops.sum_squares(self._stacked_tables(alias_ok=True), self.reg_weight)
        return reg_params(self, self.reg_weight)

    def _propagate_sum(self, adj, embeds, noises=None, eps=0.0, reg_weight=None):
        """embeds + sum_{l=1..L} P_l(adj^l embeds): one fused SpMM launch per layer.  With reg_weight: returns (sum, reg) where
        reg = reg_weight * |embeds|^2 (reg_params of the two tables, loss_utils.py:20-24) rides on the same autograd node -- its
        gradient is added in the epilogue of the last backward product instead of by kernels of its own"""
        if reg_weight is not None:
            if self._hook_overridden():
                return self._propagate_sum(adj, embeds, noises, eps), ops.sum_squares(embeds, reg_weight)
            return ops.propagate_sum(adj, embeds, self.layer_num, noises, eps, reg_weight=reg_weight)
        if self._hook_overridden():      # plugin semantics of the reference: lightgcn.py:38-41 / simgcl.py:23-29
            total, x = embeds, embeds
            for l in range(self.layer_num):
                x = self._propagate(adj, x)
                if noises is not None:
                    nz = noises[l] if t.is_tensor(noises[l]) else noises[l].materialize()      # rng.PhiloxNoise token (model.device_rng)
                    x = x + t.nn.functional.normalize(nz, p=2, dim=1) * t.sign(x) * eps
                total = total + x
            return total
        return ops.propagate_sum(adj, embeds, self.layer_num, noises, eps)

    def _split(self, embeds):
        return embeds[:self.user_num], embeds[self.user_num:]

    def _cached(self):
        """the embeddings computed by the last forward, reused across evaluation batches"""
        if not self.is_training and self.final_embeds is not None:
            return self._split(self.final_embeds)
        return None

    def _embeddings_for_eval(self):
        """propagated tables for scoring; models override `forward` signatures, so go through
        full_predict's own call convention"""
        raise NotImplementedError

    # -- all-rank scoring --------------------------------------------------------------------
    def _score_all_items(self, user_embeds, item_embeds, batch_data):
        pck_users, train_mask = batch_data
        if user_embeds.is_cuda and user_embeds.shape[1] <= ops.INFONCE_DIMS[-1] and type(self)._mask_predict is BaseModel._mask_predict:
            # scores + `_mask_predict` in one fused pass (sslrec_full_predict_f32); a subclass that overrides _mask_predict keeps its own
            return ops.full_predict(user_embeds, item_embeds, pck_users.long(), train_mask)
        scores = user_embeds[pck_users.long()] @ item_embeds.T
        return self._mask_predict(scores, train_mask)

    def predict_topk(self, users, k, trn_csr_device):
        """Top-k unseen items for `users` without the dense [B, I] train mask the reference ships
        from the host for every test batch (trainer/metrics.py:99-100: 750 MB per batch at
        amazon-book size).  `trn_csr_device` = (rowptr int64 [U+1], col int64 [nnz]) of the train
        interactions on the device; seen items get the same -1e8 offset as `_mask_predict`."""
        user_embeds, item_embeds = self._embeddings_for_eval()
        if user_embeds.shape[1] <= ops.INFONCE_DIMS[-1] and int(k) <= ops.EVAL_KMAX:      # fused MFMA tiles + CSR membership + top-k, no [B, I] matrix
            return ops.eval_topk(user_embeds, item_embeds, users.long(), k, trn_csr_device)      # (other sizes are zero-padded to a kernel width)
        # embedding sizes / k beyond the kernel's buffers: the reference expression on the device, at most EVAL_DENSE_CHUNK users at a
        # time -- Metric hands over up to 65,536 users per call, and a [65536, I] score matrix is 24 GB at amazon-book size
        users = users.long()
        rowptr, col = trn_csr_device
        out = []
        for lo in range(0, users.numel(), EVAL_DENSE_CHUNK):
            us = users[lo:lo + EVAL_DENSE_CHUNK]
            scores = user_embeds[us] @ item_embeds.T
            start, end = rowptr[us], rowptr[us + 1]
            counts = end - start
            owner = t.repeat_interleave(t.arange(us.numel(), device=us.device), counts)
            offs = t.arange(int(counts.sum()), device=us.device) - t.repeat_interleave(counts.cumsum(0) - counts, counts)
            seen = col[t.repeat_interleave(start, counts) + offs]
            scores[owner, seen] = scores[owner, seen] * 0 - 1e8
            out.append(t.topk(scores, k=k)[1])
        return t.cat(out) if out else t.empty((0, int(k)), dtype=t.int64, device=users.device)
