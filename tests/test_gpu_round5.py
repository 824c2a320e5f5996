"""Round-5 additions to the GPU parity suite (through the C ABI, like tests/test_gpu_parity.py): the factorized normalization
of a layer chain on the column-swept kernel, the compacted edge-dropped views of the row-bundled layout, eight processes on one GPU.
Needs an MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_expr as R
from tests import helpers as H  # noqa: F401

pytestmark = pytest.mark.gpu

DEV = os.environ.get('SSLREC_TEST_DEVICE', 'cuda')


def _bipartite(name='tiny', seed=2023):
    from sslrec_amd.data_utils.synth import make_dataset
    trn = R.binarize_coo(make_dataset(name, seed))
    idx, vals, n = R.normalized_bipartite_coo(trn)
    return trn, idx, vals, n


def _select_width(monkeypatch, passes):
    monkeypatch.setenv('SSLREC_SPMM_SWEPT', '1')
    if passes:
        monkeypatch.setenv('SSLREC_SWEPT_WIDTH', '32')
    else:
        monkeypatch.delenv('SSLREC_SWEPT_WIDTH', raising=False)


# ------------------------------------------------------------------------------------------
# factorized normalization (sslrec_epilogue_t.row_scale / scale_flags, ABI 6)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('passes', [False, True])
@pytest.mark.parametrize('d', [32, 64, 128])
def test_scale_flags_of_the_swept_kernel_against_fp64(d, passes, monkeypatch):
    """The three flags one by one on the reference's adjacency D^-1/2 A D^-1/2 (data_handler_general_cf.py:37-51), whose values are
    r[i] * r[j]: a PATTERN launch on the scaled operand r (.) x gives A x without reading a value; SCALE_Y / SCALE_ACC hand the next
    launch its scaled operand.  Against an fp64 product of the stored fp32 values, on the graph and on an edge-dropped view."""
    from sslrec_amd import ops
    from sslrec_amd.graph import DroppedView, PropGraph
    _select_width(monkeypatch, passes and d > 32)
    trn, idx, vals, n = _bipartite()
    g = PropGraph(idx[0], idx[1], vals, (n, n), DEV)
    fact = g.factorization()
    assert fact is not None and fact[2] and fact[0].shape == (n,)
    r = fact[0]
    gen = torch.Generator().manual_seed(d)
    x = torch.randn(n, d, generator=gen)
    acc = torch.randn(n, d, generator=gen)
    keep = torch.rand(vals.shape[0], generator=gen) < 0.6
    for which, ii in (('fwd', idx), ('bwd', idx[::-1])):
        for adj, sel in ((g, np.ones(vals.shape[0], dtype=bool)), (DroppedView(g, keep), keep.numpy())):
            ref = R.spmm_fp64(np.ascontiguousarray(ii[:, sel]), vals[sel], n, x.numpy())
            xd, accd = x.to(DEV), acc.to(DEV)
            rn = r.cpu().numpy().astype(np.float64)[:, None]
            # pattern launch on the scaled operand: Y = r (.) (A x), acc_out = acc + A x
            out = torch.empty_like(accd)
            y = ops.spmm_raw(adj, xd * r[:, None], which, acc_in=accd, acc_out=out, row_scale=r,
                             scale_flags=ops.SCALE_PATTERN | ops.SCALE_Y)
            np.testing.assert_allclose(out.cpu().numpy(), acc.numpy() + ref, rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(y.cpu().numpy(), rn * ref, rtol=2e-5, atol=2e-6)
            # pattern launch without Y scaling, without accumulator
            y2 = ops.spmm_raw(adj, xd * r[:, None], which, row_scale=r, scale_flags=ops.SCALE_PATTERN)
            np.testing.assert_allclose(y2.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
            # valued launch, accumulator scaled: acc_out = r (.) (acc + A x)
            out3 = torch.empty_like(accd)
            ops.spmm_raw(adj, xd, which, acc_in=accd, acc_out=out3, want_y=False, row_scale=r, scale_flags=ops.SCALE_ACC)
            np.testing.assert_allclose(out3.cpu().numpy(), rn * (acc.numpy() + ref), rtol=2e-5, atol=2e-6)
            # valued launch with SCALE_Y only equals the plain product scaled
            y4 = ops.spmm_raw(adj, xd, which, row_scale=r, scale_flags=ops.SCALE_Y)
            np.testing.assert_allclose(y4.cpu().numpy(), rn * ref, rtol=2e-5, atol=2e-6)


def test_scale_flags_are_rejected_where_they_do_not_apply(monkeypatch):
    from sslrec_amd import ops
    from sslrec_amd.graph import PropGraph
    trn, idx, vals, n = _bipartite()
    monkeypatch.setenv('SSLREC_SPMM_SWEPT', '0')
    g = PropGraph(idx[0], idx[1], vals, (n, n), DEV)
    r = g.factorization()[0]
    x = torch.randn(n, 64, device=DEV)
    with pytest.raises(ValueError):
        ops.spmm_raw(g, x, 'fwd', row_scale=r, scale_flags=ops.SCALE_PATTERN)      # the streamed kernel reads values
    monkeypatch.setenv('SSLREC_SPMM_SWEPT', '1')
    g2 = PropGraph(idx[0], idx[1], vals, (n, n), DEV)
    with pytest.raises(ValueError):
        ops.spmm_raw(g2, x, 'fwd', row_scale=r[:-1].contiguous(), scale_flags=ops.SCALE_Y)
    # a graph whose values do not factorize is not mistaken for one that does
    rng = np.random.default_rng(0)
    g3 = PropGraph(idx[0], idx[1], rng.uniform(0.1, 1.0, vals.shape[0]).astype(np.float32), (n, n), DEV)
    assert g3.factorization() is None
    assert ops._chain_scale(g3, 64, 3) is None and ops._chain_scale(g2, 64, 3) is not None and ops._chain_scale(g2, 64, 1) is None


@pytest.mark.parametrize('passes', [False, True])
@pytest.mark.parametrize('L', [2, 3])
@pytest.mark.parametrize('d', [32, 64])
def test_factorized_layer_chain_equals_the_valued_chain_and_the_oracle(d, L, passes, monkeypatch):
    """propagate_sum forward + backward (lightgcn.py:31-43 and its autograd) with the factorized chain (default) against the chain
    that reads the value stream in every launch (SSLREC_SPMM_FACTORIZED=0, rounds 1-4) and against the oracle: plain graph, the
    reference's edge-dropped view, with the regularizer's gradient riding on the last backward product.  North star: 1e-5 on embeddings."""
    from sslrec_amd import ops
    from sslrec_amd.graph import DroppedView, PropGraph
    _select_width(monkeypatch, passes and d > 32)
    trn, idx, vals, n = _bipartite()
    n_user = trn.shape[0]
    adj_t = R.torch_adj_from(idx, vals, n)
    gen = torch.Generator().manual_seed(7 * d + L)
    e0 = (torch.rand(n, d, generator=gen) - 0.5)
    gt = torch.randn(n, d, generator=gen) * 1e-2
    draw = torch.rand(vals.shape[0], generator=gen)
    g = PropGraph(idx[0], idx[1], vals, (n, n), DEV)
    for keep_rate in (1.0, 0.5):
        # oracle
        eo = e0.clone().requires_grad_(True)
        u, i = R.lightgcn_forward(adj_t, eo[:n_user], eo[n_user:], L, keep_rate, draw if keep_rate != 1.0 else None)
        tot_ref = torch.cat([u, i])
        reg_ref = 1e-3 * eo.square().sum()
        ((tot_ref * gt).sum() + 2.0 * reg_ref).backward()
        adj = g if keep_rate == 1.0 else DroppedView(g, R.edge_drop_mask(draw, keep_rate))
        res = {}
        for fac in (True, False):
            monkeypatch.setattr(ops, 'FACTORIZED', 1 if fac else 0)
            assert (ops._chain_scale(adj, d, L) is not None) == fac and ops._chain_scale(adj, d, L, perturbed=True) is None
            ed = e0.to(DEV).requires_grad_(True)
            tot, reg = ops.propagate_sum(adj, ed, L, reg_weight=1e-3)
            ((tot * gt.to(DEV)).sum() + 2.0 * reg).backward()
            res[fac] = (tot.detach().cpu().numpy(), ed.grad.cpu().numpy(), float(reg))
        for fac in (True, False):
            np.testing.assert_allclose(res[fac][0], tot_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(res[fac][1], eo.grad.numpy(), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(res[fac][2], float(reg_ref), rtol=1e-5)
        # the two chains agree far inside the tolerance (one extra rounding of r[i] * r[j] per entry)
        np.testing.assert_allclose(res[True][0], res[False][0], rtol=0, atol=2e-6)
        np.testing.assert_allclose(res[True][1], res[False][1], rtol=0, atol=2e-7)


@pytest.mark.parametrize('L', [2, 3])
def test_factorized_chain_under_simgcl_views_equals_the_valued_chain(L, monkeypatch):
    """SimGCL's three views (simgcl.py:39-43: two perturbed, one clean) share the first product (K epilogues) and the backward chain;
    factorized, the shared launch writes the views' SCALED tables and the later launches are pattern launches whose perturbation acts on
    the unscaled row (aug_utils.py:125-132).  Opt-in (SSLREC_SPMM_FACTORIZED=2): by default a perturbed chain keeps the valued form, whose
    sums are the reference's own fp32 products in the reference's order -- sign(y) is discontinuous and a differently rounded y ~ 0 flips it."""
    from sslrec_amd import ops
    from sslrec_amd.graph import PropGraph
    monkeypatch.setenv('SSLREC_SPMM_SWEPT', '1')
    monkeypatch.delenv('SSLREC_SWEPT_WIDTH', raising=False)
    trn, idx, vals, n = _bipartite()
    d, eps = 64, 0.9
    gen = torch.Generator().manual_seed(L)
    e0 = torch.rand(n, d, generator=gen) - 0.5
    noises = [[torch.rand(n, d, generator=gen).to(DEV) for _ in range(L)] for _ in range(2)] + [None]
    gts = [torch.randn(n, d, generator=gen).to(DEV) * 1e-2 for _ in range(3)]
    g = PropGraph(idx[0], idx[1], vals, (n, n), DEV)
    res = {}
    for fac in (True, False):
        monkeypatch.setattr(ops, 'FACTORIZED', 2 if fac else 0)      # (level 2: perturbed chains too -- not the default, see ops.FACTORIZED)
        ed = e0.to(DEV).requires_grad_(True)
        outs = ops.propagate_sum_views(g, ed, L, noises, eps)
        sum((o * t).sum() for o, t in zip(outs, gts)).backward()
        res[fac] = [o.detach().cpu().numpy() for o in outs] + [ed.grad.cpu().numpy()]
    for a, b in zip(res[True][:3], res[False][:3]):
        np.testing.assert_allclose(a, b, rtol=0, atol=3e-6)
    np.testing.assert_allclose(res[True][3], res[False][3], rtol=0, atol=3e-7)
    # and against the oracle's perturbed forward (view 0)
    adj_t = R.torch_adj_from(idx, vals, n)
    n_user = trn.shape[0]
    u, i = R.lightgcn_forward(adj_t, e0[:n_user], e0[n_user:], L, noise_draws=[z.cpu() for z in noises[0]], eps=eps)
    np.testing.assert_allclose(res[True][0], torch.cat([u, i]).numpy(), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------
# SimGCL / SGL: the step as one autograd node (ops.contrastive_step)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('model_name', ['simgcl', 'sgl'])
@pytest.mark.parametrize('case,d,L', [('tiny', 64, 3), ('tiny', 32, 2), ('yelp', 64, 2)])
def test_one_node_contrastive_step_equals_the_separate_nodes_and_the_reference(model_name, case, d, L, monkeypatch):
    """cal_loss + backward of SimGCL (simgcl.py:39-55) and SGL (sgl.py:45-65) as ONE autograd node with a hand-written backward
    (ops.contrastive_step: one gradient table for SimGCL's three views, the regularizer's gradient in the last flush, no stock launch
    in between) against the step composed from the separate nodes (SSLREC_ONE_NODE_STEP=0) on the reference's own draws, and both
    against the real reference's recorded step (losses rtol 1e-5, gradients rtol 1e-4 / atol 1e-7)."""
    import sslrec_amd.models.aug_utils as aug
    from sslrec_amd import ops
    from tests.test_gpu_parity import _check_step
    g, cfg = H.load_golden(case, model_name, d, L)
    res = {}
    for one in (True, False):
        monkeypatch.setattr(ops, 'ONE_NODE_STEP', one)
        dh, model = H.setup_model(model_name, g, cfg, DEV, d, L)
        if case == 'tiny':
            H.set_params_from_golden(model, g)
            monkeypatch.setattr(aug.t, 'rand', H.ReplayRand(H.golden_draws(g)))
        else:
            H.set_params_seeded_fill(model)
            torch.manual_seed(2023 + 1)
        seen = []
        orig = ops._ContrastiveStepFn.apply
        monkeypatch.setattr(ops, 'contrastive_step', lambda *a: (seen.append(1), orig(*a))[1])
        loss, parts = model.cal_loss(H.batch_from_golden(g, DEV))
        loss.backward()
        assert bool(seen) == one                      # the path under test really ran
        _check_step(g, model, loss, parts, full=(case == 'tiny'))
        res[one] = (loss.item(), {k: float(v) for k, v in parts.items()}, [p.grad.clone() for p in model.parameters()])
        monkeypatch.undo()
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=2e-6)
    for k in res[True][1]:
        np.testing.assert_allclose(res[True][1][k], res[False][1][k], rtol=2e-6)
    for a, b in zip(res[True][2], res[False][2]):
        scale = float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale)


# ------------------------------------------------------------------------------------------
# compacted edge-dropped views on the row-bundled layout (sslrec_bundled_compact)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('seg_max', [None, 8])
@pytest.mark.parametrize('d', [8, 16, 32])
def test_bundled_compact_view_has_the_bits_of_the_zero_valued_view_and_runs_shorter_streams(d, seg_max, monkeypatch):
    """EdgeDrop (aug_utils.py:18-31) on the row-bundled layout, compacted: every row keeps its kept entries in order, so the product --
    forward and transposed, given mask and Philox mask -- equals the zero-valued form (sslrec_bundled_drop_values, rounds 3-4) BIT FOR
    BIT and the fp64 product over the kept entries to rounding, while the streams shrink with the keep rate; keep-all / drop-all masks;
    rows cut into chunks (seg_max 8) and a heavy row included."""
    from sslrec_amd import ops
    from sslrec_amd.graph import BundledLayout, DroppedView, PropGraph
    from sslrec_amd.rng import PhiloxState
    from tests.test_gpu_parity import _rand_graph
    monkeypatch.setenv('SSLREC_SPMM_SWEPT', '0')
    if d == 32:
        monkeypatch.setenv('SSLREC_SPMM_BUNDLED32', '1')
    n_rows, n_cols = 2111, 1733
    rows, cols, vals = _rand_graph(n_rows, n_cols, 60000, seed=d, heavy_row=5)
    g = PropGraph(rows, cols, vals, (n_rows, n_cols), DEV, seg_max=seg_max)
    lay = g.fwd.packed(d)
    assert isinstance(lay, BundledLayout)
    gen = torch.Generator().manual_seed(d)
    x = torch.randn(n_cols, d, generator=gen).to(DEV)
    z = torch.randn(n_rows, d, generator=gen).to(DEV)
    st = PhiloxState(DEV, seed=3)
    st.advance()
    keep = torch.rand(vals.size, generator=gen) < 0.5
    masks = [('given', lambda: DroppedView(g, keep)), ('philox', lambda: DroppedView(g, None, 1.0, philox=(st, 2, 0.3))),
             ('all', lambda: DroppedView(g, torch.ones(vals.size, dtype=torch.bool))), ('none', lambda: DroppedView(g, torch.zeros(vals.size, dtype=torch.bool)))]
    for name, make in masks:
        outs = {}
        for compacted in (True, False):
            monkeypatch.setenv('SSLREC_BUNDLED_COMPACT', '1' if compacted else '0')
            view = make()
            outs[compacted] = (ops.spmm_raw(view, x, 'fwd'), ops.spmm_raw(view, z, 'bwd'))
            if compacted:
                col, val, b_steps, w_blocks = view.compact('fwd', d)
                assert col is not None and b_steps.numel() == lay.n_bundles and w_blocks.numel() == lay.n_waves
                used, total = int(w_blocks.sum().item()) * 64, lay.n_elem
                kept_frac = {'given': 0.5, 'philox': 0.3, 'all': 1.0, 'none': 0.0}[name]
                assert used <= total and (name != 'all' or used == total)
                if name in ('given', 'philox') and seg_max is None:
                    # the longest of a bundle's rows sets its length: somewhat above the keep rate.  (seg_max 8 cuts the rows into chunks of
                    # at most 8 entries -- one or two blocks of S steps each -- whose bundles cannot get shorter than a block.)
                    assert kept_frac * 0.9 < used / total < kept_frac + 0.25, (name, used / total)
                if name == 'none':
                    assert used == 0
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1]), name
        if name == 'given':
            kn = keep.numpy()
            np.testing.assert_allclose(outs[True][0].cpu().numpy(), R.spmm_fp64(np.vstack([rows[kn], cols[kn]]), vals[kn], n_rows, x.cpu().numpy()),
                                       rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(outs[True][1].cpu().numpy(), R.spmm_fp64(np.vstack([cols[kn], rows[kn]]), vals[kn], n_cols, z.cpu().numpy()),
                                       rtol=1e-5, atol=1e-5)
        if name == 'none':
            assert not outs[True][0].any() and not outs[True][1].any()


# ------------------------------------------------------------------------------------------
# the one-launch reductions' ordering is a hardware property -- hammer it
# ------------------------------------------------------------------------------------------
def test_one_launch_reductions_survive_thousands_of_back_to_back_launches():
    """sslrec_sumsq_fwd_f32 / sslrec_bpr_fwd_f32 finish in ONE launch: the last workgroup adds the partials the others published with a
    returning device-scope atomic exchange (csrc/losses.hip: finish_by_last_block -- no release / acquire pair, gfx942 / gfx950 complete
    such atomics at the coherence point).  4,000 back-to-back launches on alternating inputs, every result compared with the value the
    same input gave the first time: a stale partial (or a ticket counter left non-zero) would show as a different sum."""
    from sslrec_amd import ops
    gen = torch.Generator().manual_seed(1)
    xs = [torch.randn(144242, 64, generator=gen).to(DEV) * s for s in (0.1, 0.3)]
    want = [float(x.double().square().sum()) for x in xs]
    outs = []
    with torch.no_grad():
        for it in range(4000):
            outs.append(ops.sum_squares(xs[it & 1], 1.0))
    got = torch.stack(outs).cpu().numpy()
    first = [got[0], got[1]]
    assert all(got[k] == first[k & 1] for k in range(len(got)))          # bit-repeatable: the partials are added in a fixed order
    np.testing.assert_allclose(first, want, rtol=2e-6)
    n_user, n_item, B, d = 5000, 7000, 4096, 64
    tab = torch.randn(n_user + n_item, d, generator=gen).to(DEV) * 0.2
    idx = [torch.randint(0, n_user, (B,), generator=gen).to(DEV), torch.randint(0, n_item, (B,), generator=gen).to(DEV),
           torch.randint(0, n_item, (B,), generator=gen).to(DEV)]
    with torch.no_grad():
        b = torch.stack([ops.bpr_loss_stacked(tab, n_user, *idx, divisor=B) for _ in range(2000)]).cpu().numpy()
    assert (b == b[0]).all()
    ref = R.cal_bpr_loss(tab[:n_user][idx[0]].cpu(), tab[n_user:][idx[1]].cpu(), tab[n_user:][idx[2]].cpu()) / B
    np.testing.assert_allclose(b[0], ref.item(), rtol=1e-5)


def _check_topk_lists(got_idx, got_val, ue, ie, dense, k, tie_user=2):
    """fused-evaluation lists against the fp64 expression of full_predict + _mask_predict + topk, all users at once (per-user Python
    loops of small torch ops cost minutes late in the suite): list lengths and -1 padding, no train item, scores to 1e-5, the same items
    wherever the k-th and (k+1)-th scores are further apart than the arithmetic's error, and for the all-scores-equal user the first k
    unseen ids in order.  Returns the number of users whose item sets were compared."""
    U = ue.shape[0]
    ref = ue.double() @ ie.double().T
    ref[dense] = -float('inf')
    ref_val, ref_idx = torch.topk(ref, k + 1)
    m = torch.clamp((~dense).sum(1), max=k)                             # list length per user
    filled = torch.arange(k)[None, :] < m[:, None]
    assert (got_idx[~filled] == -1).all() and (got_idx[filled] >= 0).all()
    rows = torch.arange(U)[:, None].expand(U, k)
    assert not dense[rows[filled], got_idx[filled]].any()               # never a train item
    assert got_idx[tie_user].tolist() == (~dense[tie_user]).nonzero()[:k, 0].tolist()
    others = filled.clone()
    others[tie_user] = False
    np.testing.assert_allclose(got_val[others].double().numpy(), ref_val[:, :k][others].numpy(), rtol=1e-5, atol=2e-6)
    apart = (m == k) & (ref_val[:, k - 1] - ref_val[:, k] > 1e-5)
    apart[tie_user] = False
    assert torch.equal(torch.sort(got_idx[apart], 1)[0], torch.sort(ref_idx[apart, :k], 1)[0])
    return int(apart.sum())


@pytest.mark.parametrize('d,k', [(64, 1), (64, 10), (64, 40), (64, 64), (32, 40)])
def test_evaluation_with_many_item_splits_bounds_the_kth_best_by_the_splits_best_scores(d, k):
    """a few hundred users against 40,000 items: 64 item splits per user group, each publishing the best score it has seen; the k-th
    largest of those is the threshold every split adopts (csrc/eval.hip `share`, n_split >= k).  The lists must still be EXACTLY the
    reference's `full_predict` + `_mask_predict` + topk (lightgcn.py:58-66, base_model.py:35-36, metrics.py:99-108): scores arriving in
    ascending / descending order, a user whose best items all sit in one split (the bound is weak there, never wrong), all scores
    equal (the first k unseen ids), a user with fewer unseen items than k, train items masked"""
    from sslrec_amd import ops
    gen = torch.Generator().manual_seed(100 * d + k)
    U, I = 300, 40000
    ue, ie = torch.randn(U, d, generator=gen) * 0.1, torch.randn(I, d, generator=gen) * 0.1
    ie[:600] += 0.05 * ue[3] / ue[3].norm()                             # user 3: the best items are the first split's
    order = torch.argsort(ie @ ue[0])
    ie = ie[order].contiguous()                                         # ascending for user 0
    ue[1] = -ue[0]                                                      # descending for user 1
    ue[2] = 0.0                                                         # all scores equal
    dense = torch.rand(U, I, generator=gen) < 0.02
    dense[5] = True
    dense[5, torch.randperm(I, generator=gen)[:7]] = False              # user 5: 7 unseen items
    rowptr = torch.zeros(U + 1, dtype=torch.int64)
    rowptr[1:] = dense.sum(1).cumsum(0)
    col = dense.nonzero()[:, 1].contiguous()
    got_idx, got_val = ops.eval_topk(ue.to(DEV), ie.to(DEV), None, k, (rowptr.to(DEV), col.to(DEV)), return_scores=True)
    assert _check_topk_lists(got_idx.cpu(), got_val.cpu(), ue, ie, dense, k) > 0.5 * U


@pytest.mark.parametrize('U,I,d,k,what', [
    (300, 2100, 64, 40, 'exact-fp32 tiles, 4 item splits: every cut publishes the split\'s 10th best, the bound is the 4th largest'),
    (2500, 5000, 64, 40, 'fp16-plane tiles (>= 2048 users), 9 splits: the shared maximum of the splits\' own k-th bests only'),
    (16300, 3000, 64, 40, 'fp16-plane tiles, 4 splits, m = 10'),
    (16300, 3000, 128, 20, 'd = 128, fp16-plane tiles, 4 splits, m = 5'),
    (16300, 3000, 32, 5, 'd = 32, 4 splits, m = 2, j = 3'),
])
@pytest.mark.parametrize('share_few', ['0', '1'])
def test_evaluation_with_few_item_splits_bounds_the_kth_best_by_the_splits_mth_bests(U, I, d, k, what, share_few, monkeypatch):
    """the all-users shape of the fused evaluation (three or four item splits per user group): every cut of a split's buffer publishes its
    m-th best score, m = ceil(k / n_split), and the j-th largest of the published values (j m >= k distinct items at or above it) bounds
    the user's k-th best (csrc/eval.hip `share_few`); from 2048 users on the score tiles run on two fp16 planes per table.  The lists
    against the fp64 expression of `full_predict` + `_mask_predict` + topk (lightgcn.py:58-66, base_model.py:35-36, metrics.py:99-108):
    same items wherever the k-th and (k+1)-th scores are further apart than the arithmetic's error, scores to 1e-5, no train item, users
    with fewer than k unseen items padded with -1, ascending / descending arrival, all scores equal.  The m-th-best bound is opt-in
    (SSLREC_EVAL_SHARE_FEW=1, read per call: measured no faster for all amazon-book users); both forms are held to the same lists"""
    from sslrec_amd import ops
    monkeypatch.setenv('SSLREC_EVAL_SHARE_FEW', share_few)
    gen = torch.Generator().manual_seed(U + I + d + k)
    ue, ie = torch.randn(U, d, generator=gen) * 0.1, torch.randn(I, d, generator=gen) * 0.1
    order = torch.argsort(ie @ ue[0])
    ie = ie[order].contiguous()
    ue[1] = -ue[0]
    ue[2] = 0.0
    dense = torch.rand(U, I, generator=gen) < 0.02
    dense[5] = True
    dense[5, torch.randperm(I, generator=gen)[:7]] = False
    rowptr = torch.zeros(U + 1, dtype=torch.int64)
    rowptr[1:] = dense.sum(1).cumsum(0)
    col = dense.nonzero()[:, 1].contiguous()
    got_idx, got_val = ops.eval_topk(ue.to(DEV), ie.to(DEV), None, k, (rowptr.to(DEV), col.to(DEV)), return_scores=True)
    assert _check_topk_lists(got_idx.cpu(), got_val.cpu(), ue, ie, dense, k) > 0.5 * U
