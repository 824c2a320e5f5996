"""Fused Adam for the embedding tables (opt-in `optimizer.fused: true`).

Same update as `torch.optim.Adam(params, lr, weight_decay)` with the default betas/eps -- what the
reference's Trainer constructs (trainer/trainer.py:45-49) -- as ONE HIP pass per tensor
(`sslrec_adam_apply_f32`) instead of PyTorch's ~10 multi-tensor launches; the step count and the bias
corrections live on the device (`sslrec_adam_tick`), so the step can sit inside a captured hipGraph."""
import torch

from . import _lib, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._ticks = {}            # per parameter group: the device-side step counter + bias corrections

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group['betas']
            live = [p for p in group['params'] if p.grad is not None]
            if not live:
                continue
            ops._need_gpu(*live)
            if gi not in self._ticks:
                self._ticks[gi] = torch.zeros(4, dtype=torch.float32, device=live[0].device)
            tick = self._ticks[gi]
            st = ops._stream()
            _lib.check(lib.sslrec_adam_tick(tick.data_ptr(), float(group['lr']), float(b1), float(b2), st), 'sslrec_adam_tick')
            for p in live:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError('FusedAdam handles contiguous fp32 parameters')
                state = self.state[p]
                if not state:
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                rc = lib.sslrec_adam_apply_f32(p.data_ptr(), g.data_ptr(), state['exp_avg'].data_ptr(),
                                               state['exp_avg_sq'].data_ptr(), p.numel(), tick.data_ptr(), float(b1),
                                               float(b2), float(group['eps']), float(group['weight_decay']), st)
                _lib.check(rc, 'sslrec_adam_apply_f32')
        return None
