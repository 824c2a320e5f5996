"""The fused InfoNCE of BASELINE cfg 3's item term (B = 4096 anchors against 91,599 rows, d = 64, temp 0.2) in every precision
mode: forward / forward+backward time, and the error of the loss and of both gradients against the reference expression
(loss_utils.py:30-39) evaluated in float64 on the device in anchor chunks.   usage: python tools/infonce_modes.py [out.json]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
dev = 'cuda:0'
n_item, d, B, temp = 91599, 64, 4096, 0.2
gen = torch.Generator().manual_seed(5)
t1 = (torch.randn(n_item, d, generator=gen) * 0.1).to(dev)
t2 = (torch.randn(n_item, d, generator=gen) * 0.1).to(dev)
idx = torch.randint(0, n_item, (B,), generator=gen).to(dev)


def ref64():
    a, b = t1.double().requires_grad_(True), t2.double().requires_grad_(True)
    tot = 0.0
    nrm = lambda x: x / torch.sqrt(1e-8 + x.square().sum(-1, keepdim=True))
    for lo in range(0, B, 256):
        ii = idx[lo:lo + 256]
        e1, e2, al = nrm(a[ii]), nrm(b[ii]), nrm(b)
        nume = -(e1 * e2 / temp).sum(-1)
        deno = torch.log(torch.exp(e1 @ al.T / temp).sum(-1))
        part = (nume + deno).sum()
        (part / B).backward()
        tot += part.item()
    return tot, a.grad, b.grad


want, g1, g2 = ref64()
out = {}
MODES = os.environ.get('INFONCE_MODES', 'fp32,x6,h3,x6a,x63,x36,x3').split(',')
# every mode twice: the round-3 three-pass form (row-sum forward; both roles backward) and the round-4 default, in which a
# differentiated forward keeps the anchor-gradient sums (SSLREC_INFONCE_FWD_W)
for mode, fwd_w in [(m, w) for m in MODES for w in (False, True)]:
    ops.INFONCE_FWD_W = fwd_w
    a, b = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
    loss = ops.infonce_loss_gathered(a, b, idx, temp, precision=mode)
    (loss / B).backward()
    rec = {'loss_rel_err': abs(loss.item() - want) / abs(want)}
    for name, got, ref in (('grad_anchor_table', a.grad, g1), ('grad_all_table', b.grad, g2)):
        err = (got.double() - ref).abs()
        rec[name + '_max_abs_err_over_max_abs'] = (err.max() / ref.abs().max()).item()
        rec[name + '_rms_err_over_rms'] = (err.square().mean().sqrt() / ref.square().mean().sqrt()).item()
    def fwd():
        return ops.infonce_loss_gathered(a, b, idx, temp, precision=mode)
    def fwd_nograd():
        with torch.no_grad():
            return ops.infonce_loss_gathered(a, b, idx, temp, precision=mode)
    def fb():
        a.grad = b.grad = None
        (ops.infonce_loss_gathered(a, b, idx, temp, precision=mode) / B).backward()
    for nm, fn in (('fwd_nograd_ms', fwd_nograd), ('fwd_differentiable_ms', fwd), ('fwdbwd_ms', fb)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        rec[nm] = e0.elapsed_time(e1) / 10
    out[mode + ('' if fwd_w else '_three_pass')] = rec
    print(mode, 'fwd_w' if fwd_w else 'three-pass', {k: (round(v, 4) if k.endswith('_ms') else float('%.3g' % v)) for k, v in rec.items()}, flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
