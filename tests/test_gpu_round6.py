"""Round-6 GPU parity tests (all through the C ABI): the InfoNCE hot loop in its software-pipelined and round-5 orders of work, the
folded preparation / finishing launches, h3 across the temperature range of the reference's tuner (simgcl.yml: tune.temperature
reaches 1.0; below 0.0931 h3 runs as x6), and the evaluation metrics at amazon-book size."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_expr as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ref64(e1, e2, al, temp):
    """cal_infonce_loss (loss_utils.py:30-39) in float64 with its three gradients"""
    # (float64 ON THE DEVICE: the host of the GPU box has 256 cores, and torch's CPU kernels for these small shapes take ~10 s there)
    a, b, c = (x.to(DEV).double().requires_grad_(True) for x in (e1, e2, al))
    nrm = lambda x: x / torch.sqrt(1e-8 + x.square().sum(-1, keepdim=True))
    n1, n2, na = nrm(a), nrm(b), nrm(c)
    loss = (-(n1 * n2 / temp).sum(-1) + torch.log(torch.exp(n1 @ na.T / temp).sum(-1))).sum()
    loss.backward()
    return loss.item(), a.grad.cpu(), b.grad.cpu(), c.grad.cpu()


@pytest.mark.parametrize('temp', [0.05, 0.0931, 0.1, 0.5, 1.0])
@pytest.mark.parametrize('d', [32, 64, 128])
def test_h3_infonce_across_the_tuner_temperature_range(d, temp, monkeypatch):
    """The default arithmetic (h3: two fp16 planes) at the ends of the temperature range: tau = 1.0 (simgcl.yml tune.temperature)
    where the exponent bias saturates at 7, and tau < log2(e) / 15.5 where the bias would be negative and the call runs x6 instead
    (the gradients degraded silently there).  Loss and all three gradients against the reference expression in float64,
    at the tolerances of the exact-fp32 kernels."""
    from sslrec_amd import ops
    monkeypatch.delenv('SSLREC_INFONCE_PRECISION', raising=False)
    B, M = 515, 2077
    gen = torch.Generator().manual_seed(d + int(temp * 1000))
    e1, e2, al = (torch.randn(n, d, generator=gen) for n in (B, B, M))
    want, g1, g2, g3 = _ref64(e1, e2, al, temp)
    for precision in (None, 'h3'):        # the process default, and h3 asked for by name
        a, b, c = (x.to(DEV).requires_grad_(True) for x in (e1, e2, al))
        out = ops.infonce_loss(a, b, c, temp, precision=precision)
        np.testing.assert_allclose(out.item(), want, rtol=1e-5)
        out.backward()
        for got, ref in ((a, g1), (b, g2), (c, g3)):
            scale = ref.abs().max().item()
            np.testing.assert_allclose(got.grad.cpu().double().numpy(), ref.numpy(), rtol=2e-4, atol=2e-6 * scale)


@pytest.mark.parametrize('pipe,fold', [('1', '1'), ('0', '1'), ('1', '0'), ('0', '0')])
def test_infonce_loop_orders_and_folded_launches_agree(pipe, fold):
    """SSLREC_INFONCE_PIPE (the software-pipelined hot loop, three stage buffers) and SSLREC_INFONCE_FOLD (row-normalization backward
    in the all-gradient role's epilogue) are read once per process: every combination runs in a process of its own and must give the
    same loss and gradients as the reference expression -- and the two loop orders the SAME BITS (the arithmetic per tile is
    identical, only its place in the instruction stream moves)."""
    import subprocess, sys, json, tempfile
    code = r'''
import sys, json, hashlib, torch
sys.path.insert(0, %r)
from sslrec_amd import ops
out = {}
for prec in ('h3', 'x6'):
    for d, B, M, temp in ((64, 515, 2077, 0.2), (64, 1024, 3001, 0.2), (128, 300, 1500, 0.5), (32, 4096, 9000, 0.2)):
        gen = torch.Generator().manual_seed(d + B)
        t1 = torch.randn(M, d, generator=gen).cuda().requires_grad_(True)
        t2 = torch.randn(M, d, generator=gen).cuda().requires_grad_(True)
        idx = torch.randint(0, M, (B,), generator=gen).cuda()
        loss = ops.infonce_loss_gathered(t1, t2, idx, temp, precision=prec)
        loss.backward()
        h = hashlib.sha256(); h.update(t1.grad.cpu().numpy().tobytes()); h.update(t2.grad.cpu().numpy().tobytes())
        out['%%s_%%d_%%d_%%d' %% (prec, d, B, M)] = [loss.item(), h.hexdigest(), float(t1.grad.abs().sum()), float(t2.grad.abs().sum())]
json.dump(out, open(sys.argv[1], 'w'))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(p, f):
        with tempfile.NamedTemporaryFile(suffix='.json') as tf:
            env = dict(os.environ, SSLREC_INFONCE_PIPE=p, SSLREC_INFONCE_FOLD=f)
            env.pop('SSLREC_INFONCE_PRECISION', None)
            subprocess.run([sys.executable, '-c', code, tf.name], env=env, check=True)
            return json.load(open(tf.name))
    got = run(pipe, fold)
    base = run('0', '0')
    for k, (loss, sha, s1, s2) in got.items():
        bl, bsha, b1, b2 = base[k]
        np.testing.assert_allclose(loss, bl, rtol=1e-6)
        np.testing.assert_allclose([s1, s2], [b1, b2], rtol=1e-5)
        if fold == '0':
            assert loss == bl and sha == bsha, 'the loop order changed bits: %s' % k


def test_infonce_forward_loss_is_reproducible_over_many_one_launch_finishes():
    """the forward finish adds the per-workgroup partial losses in the workgroup that finishes last (tickets in the workspace, zeroed by
    the call's preparation launch): thousands of back-to-back calls on fresh and on re-used workspaces give one value"""
    from sslrec_amd import ops
    gen = torch.Generator().manual_seed(3)
    t1 = torch.randn(5000, 64, generator=gen).to(DEV)
    t2 = torch.randn(5000, 64, generator=gen).to(DEV)
    idx = torch.randint(0, 5000, (2048,), generator=gen).to(DEV)
    with torch.no_grad():
        first = ops.infonce_loss_gathered(t1, t2, idx, 0.2).item()
        vals = torch.stack([ops.infonce_loss_gathered(t1, t2, idx, 0.2) for _ in range(2000)])
    assert torch.all(vals == first), (first, vals.min().item(), vals.max().item())
    a = t1.clone().requires_grad_(True)
    b = t2.clone().requires_grad_(True)
    vals = []
    for _ in range(300):
        a.grad = b.grad = None
        loss = ops.infonce_loss_gathered(a, b, idx, 0.2)
        loss.backward()
        vals.append(loss.detach())
    vals = torch.stack(vals)
    assert torch.all(vals == vals[0])
    np.testing.assert_allclose(vals[0].item(), first, rtol=1e-6)


def test_metric_eval_at_amazon_book_size_equals_the_dense_fp32_path():
    """`Metric.eval`'s numbers (recall / ndcg / precision / mrr @ 10 / 20 / 40, trainer/metrics.py:82-127) for ALL 52,643 users of the
    amazon-book-shaped graph through the fused top-k (fp16-plane score tiles from 2048 users on, train items masked from the device
    CSR) against the reference's own path -- `full_predict` in exact fp32 (lightgcn.py:58-66), `_mask_predict` with the dense mask,
    `torch.topk` -- on the same embeddings.  The existing equality test is tiny; a float GEMM can only swap items whose scores tie to
    ~1e-7, so the metric sums must agree far inside the third digit the reference logs, and the top-40 SETS for all but a handful of
    users."""
    import scipy.sparse as sp
    from sslrec_amd import ops
    from sslrec_amd.config.configurator import load_config
    from sslrec_amd.data_utils.synth import make_dataset, split_holdout
    from sslrec_amd.trainer.metrics import Metric
    load_config('lightgcn', device=DEV, overrides={'test': {'metrics': ['recall', 'ndcg', 'precision', 'mrr'], 'k': [10, 20, 40]}})
    trn = make_dataset('amazon-book', 2023)
    tst = split_holdout(trn, 0.02, 5)                      # ~47,000 held-out interactions
    n_user, n_item = trn.shape
    gen = torch.Generator().manual_seed(1)
    ue = (torch.randn(n_user, 64, generator=gen) * 0.1).to(DEV)
    ie = (torch.randn(n_item, 64, generator=gen) * 0.1).to(DEV)
    csr = sp.csr_matrix(trn)
    csr.sort_indices()
    rowptr = torch.from_numpy(csr.indptr.astype(np.int64)).to(DEV)
    col = torch.from_numpy(csr.indices.astype(np.int64)).to(DEV)
    per_user = [[] for _ in range(n_user)]
    for u, i in zip(tst.row.tolist(), tst.col.tolist()):
        per_user[u].append(i)
    users = np.array(sorted(set(tst.row.tolist())), dtype=np.int64)
    assert len(users) > 20000
    k = 40
    fused = ops.eval_topk(ue, ie, torch.from_numpy(users).to(DEV), k, (rowptr, col)).cpu()           # one launch: the fp16-plane tiles
    dense = []
    for lo in range(0, len(users), 1024):                                                            # the reference's batches of 1024
        us = torch.from_numpy(users[lo:lo + 1024]).to(DEV)
        mask = torch.zeros(len(us), n_item, device=DEV)
        cnt = rowptr[us + 1] - rowptr[us]
        owner = torch.repeat_interleave(torch.arange(len(us), device=DEV), cnt)
        offs = torch.arange(int(cnt.sum()), device=DEV) - torch.repeat_interleave(cnt.cumsum(0) - cnt, cnt)
        mask[owner, col[torch.repeat_interleave(rowptr[us], cnt) + offs]] = 1.0
        scores = R.full_predict(ue, ie, us, mask)                                                     # exact fp32 + _mask_predict
        dense.append(torch.topk(scores, k)[1].cpu())
    dense = torch.cat(dense)
    metric = Metric()
    truth = [per_user[u] for u in users.tolist()]
    a = metric.eval_batch((fused, truth), [10, 20, 40])
    b = metric.eval_batch((dense, truth), [10, 20, 40])
    for m in a:
        assert b[m].sum() > 0
        np.testing.assert_allclose(a[m], b[m], rtol=2e-4, err_msg=m)
    differing = sum(set(x) != set(y) for x, y in zip(fused.tolist(), dense.tolist()))
    assert differing <= len(users) // 500, differing


@pytest.mark.parametrize('d', [32, 64, 128])
@pytest.mark.parametrize('scale,temp', [(0.3, 0.1), (1.0, 0.2), (0.05, 0.05), (3.0, 1.0)])
def test_h3_on_the_unnormalized_variant_against_float64(d, scale, temp):
    """LightGCL's contrastive term (lightgcl.py:114-118: no normalization, clamped positive pair, + 1e-8 inside the log) on two fp16
    planes: plane scales from the tables' largest magnitudes and a per-anchor exponent bias from a row-max pre-pass, all chosen on the
    device (csrc/infonce.hip, dyn_*).  Rows of very different norms, scores from -43 to +43 log2 units, against the reference
    expression in float64 -- and against x6 (three bf16 planes), whose tolerances it must meet."""
    from sslrec_amd import ops
    gen = torch.Generator().manual_seed(d + int(100 * temp))
    n, B = 1501, 700
    row_scale = torch.exp(torch.randn(n, 1, generator=gen) * 0.7)                     # row norms spread over a factor of ~20
    t1 = torch.randn(n, d, generator=gen) * row_scale
    t2 = torch.randn(n, d, generator=gen) * row_scale[torch.randperm(n, generator=gen)]
    idx = torch.randint(0, n, (B,), generator=gen)
    idx[:7] = 11
    # magnitudes: `scale` sets how far the two tables' magnitudes are apart (the device chooses one power of two per table), the largest
    # |score| is brought to 30 (e^30; beyond e^88 the reference's own exp overflows -- tested below)
    t1, t2 = t1 * scale, t2 / scale
    top = (t1.to(DEV).double()[idx.to(DEV)] @ t2.to(DEV).double().T / temp).abs().max().item()
    t1 = t1 * float(np.sqrt(30.0 / top))
    t2 = t2 * float(np.sqrt(30.0 / top))
    a64, b64 = t1.to(DEV).double().requires_grad_(True), t2.to(DEV).double().requires_grad_(True)      # (float64 on the device, see _ref64)
    idx_d = idx.to(DEV)
    sc = a64[idx_d] @ b64.T / temp
    assert 25.0 < sc.abs().max().item() < 35.0
    ref = (torch.log(torch.exp(sc).sum(1) + 1e-8) - torch.clamp((a64[idx_d] * b64[idx_d]).sum(1) / temp, -5.0, 5.0)).sum()
    ref.backward()
    got = {}
    for prec in ('h3', 'x6'):
        a, b = t1.to(DEV).requires_grad_(True), t2.to(DEV).requires_grad_(True)
        out = ops.infonce_loss_gathered(a, b, idx.to(DEV), temp, variant=1, precision=prec)
        np.testing.assert_allclose(out.item(), ref.item(), rtol=1e-5, err_msg=prec)
        out.backward()
        got[prec] = (a.grad.cpu().double(), b.grad.cpu().double())
        with torch.no_grad():                                                         # the no-grad forward runs the row-sum kernel
            np.testing.assert_allclose(ops.infonce_loss_gathered(a, b, idx.to(DEV), temp, variant=1, precision=prec).item(), ref.item(), rtol=1e-5)
    for k, want in enumerate((a64.grad.cpu(), b64.grad.cpu())):
        scale_g = want.abs().max().item()
        err_h3 = (got['h3'][k] - want).abs().max().item() / scale_g
        err_x6 = (got['x6'][k] - want).abs().max().item() / scale_g
        # 22-bit operands: a score of magnitude |t| (log2 units) carries ~|t| 2^-22 of error, which is the relative error of its softmax
        # weight -- 1e-5 of the gradient's scale at |t| = 43, the largest here (x6's 24-bit operands: a quarter of that)
        assert err_h3 < 2e-5, (k, err_h3, err_x6)
        assert err_h3 < 8 * err_x6 + 2e-6, (k, err_h3, err_x6)


def test_h3_unnormalized_matches_the_reference_where_its_exp_overflows():
    """scores beyond 88.7: the reference's exp() is inf in fp32 and so is its loss; the row sums here go back to their true scale
    (x 2^-bias) and overflow at the same point -- the value is matched, not improved on"""
    from sslrec_amd import ops
    gen = torch.Generator().manual_seed(0)
    t1 = torch.randn(300, 64, generator=gen)
    t2 = torch.randn(300, 64, generator=gen)
    t2[5] = t1[9] * 3.0                                                               # <t1[9], t2[5]> / 0.1 ~ 1900
    idx = torch.tensor([9, 1, 2])
    ref = R.lightgcl_cl_terms if hasattr(R, 'lightgcl_cl_terms') else None
    want = (torch.log(torch.exp(t1[idx] @ t2.T / 0.1).sum(1) + 1e-8) - torch.clamp((t1[idx] * t2[idx]).sum(1) / 0.1, -5.0, 5.0)).sum()
    assert torch.isinf(want)
    with torch.no_grad():
        out = ops.infonce_loss_gathered(t1.to(DEV), t2.to(DEV), idx.to(DEV), 0.1, variant=1, precision='h3')
    assert torch.isinf(out) and out.item() > 0
