"""CPU oracle for the general-CF hot path  --  TEST INFRASTRUCTURE ONLY.

This module restates, in plain PyTorch/scipy on the CPU, the arithmetic that
the reference (HKUDS/SSLRec) performs on the path named by BASELINE.json:
LightGCN-style propagation over the normalized bipartite adjacency and the
BPR / InfoNCE losses.  It exists to CHECK the HIP path; nothing under
``sslrec_amd/`` may import it.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg use it.

Parity pinning: the reference ships no tests or golden vectors,
so this restatement is pinned against the reference *itself*:
``oracle/make_golden.py`` imports the real reference from ``/root/reference``
on the CPU, records inputs/outputs into ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` asserts this file reproduces them.

Every function cites the reference lines it follows (paths relative to the
reference root).
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# adjacency  (data_utils/data_handler_general_cf.py:22-73)
# --------------------------------------------------------------------------
def binarize_coo(mat):
    """`_load_one_mat` post-processing (data_handler_general_cf.py:31-35):
    non-zeros become 1.0f, result is a scipy COO matrix."""
    mat = (mat != 0).astype(np.float32)
    if not isinstance(mat, sp.coo_matrix):
        mat = sp.coo_matrix(mat)
    return mat


def normalized_bipartite_coo(trn_mat):
    """Build D^-1/2 [[0,R],[R^T,0]] D^-1/2 exactly the way the reference does
    (data_handler_general_cf.py:37-51, 53-73), including the float64 math,
    the +1e-10 in the degree and the column-major entry ORDER that falls out
    of ``mat.dot(D).transpose().dot(D).tocoo()``.

    Returns (idx int64 [2,nnz], vals float32 [nnz], N)."""
    n_user, n_item = trn_mat.shape
    ul = sp.csr_matrix((n_user, n_user))
    lr = sp.csr_matrix((n_item, n_item))
    big = sp.vstack([sp.hstack([ul, trn_mat]), sp.hstack([trn_mat.transpose(), lr])])
    big = (big != 0) * 1.0
    deg = np.array(big.sum(axis=-1)) + 1e-10
    dinv = np.reshape(np.power(deg, -0.5), [-1])
    dinv[np.isinf(dinv)] = 0.0
    dmat = sp.diags(dinv)
    norm = big.dot(dmat).transpose().dot(dmat).tocoo()
    idx = np.vstack([norm.row, norm.col]).astype(np.int64)
    vals = norm.data.astype(np.float32)
    return idx, vals, n_user + n_item


def torch_adj_from(idx, vals, n):
    """Uncoalesced sparse COO tensor as `_make_torch_adj` returns it
    (data_handler_general_cf.py:70-73)."""
    return torch.sparse_coo_tensor(torch.as_tensor(idx), torch.as_tensor(vals), (n, n))


def lightgcl_adj(trn_mat):
    """LightGCL's own U x I normalization (models/general_cf/lightgcl.py:16-22):
    data / sqrt(rowD * colD), computed per entry in the pickle's float32 dtype,
    then coalesced.  Returns a coalesced torch sparse tensor."""
    m = trn_mat.tocoo().astype(np.float32)
    row_d = np.array(m.sum(1)).squeeze()
    col_d = np.array(m.sum(0)).squeeze()
    data = m.data.copy()
    # Same per-entry scalar loop as the reference: every operand is a numpy float32
    # SCALAR there (float32 sums, float32 product, `pow(np.float32, 0.5)` stays
    # float32 under NumPy-2 promotion and goes through the scalar libm powf, which
    # differs by 1 ulp from numpy's vectorised power loop), so no vectorisation here.
    rows, cols = m.row, m.col
    for i in range(len(data)):
        data[i] = data[i] / pow(row_d[rows[i]] * col_d[cols[i]], 0.5)
    idx = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(data), m.shape).coalesce()


# --------------------------------------------------------------------------
# augmentations  (models/aug_utils.py)
# --------------------------------------------------------------------------
def edge_drop_mask(rand_draw, keep_rate):
    """aug_utils.py:28 -- `(rand + keep_rate).floor().bool()`."""
    return (rand_draw + keep_rate).floor().type(torch.bool)


def edge_drop(adj, keep_rate, rand_draw=None, resize_val=False):
    """aug_utils.py:18-31.  `rand_draw` lets a test supply the recorded draw;
    otherwise one is taken from the global CPU generator like the reference."""
    if keep_rate == 1.0:
        return adj
    vals = adj._values()
    idxs = adj._indices()
    if rand_draw is None:
        rand_draw = torch.rand(vals.size())
    mask = edge_drop_mask(rand_draw, keep_rate)
    new_vals = vals[mask] / (keep_rate if resize_val else 1.0)
    return torch.sparse_coo_tensor(idxs[:, mask], new_vals, adj.shape)


def embed_perturb(embeds, eps, rand_draw=None):
    """aug_utils.py:125-132: E + eps * sign(E) * normalize(rand, p=2, dim=1)."""
    if rand_draw is None:
        rand_draw = torch.rand(embeds.shape)
    noise = (F.normalize(rand_draw, p=2) * torch.sign(embeds)) * eps
    return embeds + noise


# --------------------------------------------------------------------------
# propagation  (models/general_cf/lightgcn.py:28-43, simgcl.py:20-30, sgl.py:20-36)
# --------------------------------------------------------------------------
def propagate(adj, embeds):
    """lightgcn.py:28-29."""
    return torch.spmm(adj, embeds)


def lightgcn_forward(adj, user_embeds, item_embeds, layer_num, keep_rate=1.0,
                     mask_draw=None, noise_draws=None, eps=None, return_layers=False):
    """Layer loop + SUM aggregation (lightgcn.py:31-43).  With `noise_draws`
    (list of L draws) it is SimGCL's perturbed forward (simgcl.py:23-30); with
    `keep_rate<1` it is the edge-dropped forward LightGCN/SGL-ED use
    (lightgcn.py:36-37, sgl.py:27-28)."""
    n_user = user_embeds.shape[0]
    embeds = torch.concat([user_embeds, item_embeds], dim=0)
    layers = [embeds]
    if keep_rate != 1.0:
        adj = edge_drop(adj, keep_rate, mask_draw)
    for l in range(layer_num):
        embeds = propagate(adj, layers[-1])
        if noise_draws is not None:
            embeds = embed_perturb(embeds, eps, noise_draws[l])
        layers.append(embeds)
    total = sum(layers)
    if return_layers:
        return total[:n_user], total[n_user:], layers
    return total[:n_user], total[n_user:]


# --------------------------------------------------------------------------
# losses  (models/loss_utils.py)
# --------------------------------------------------------------------------
def cal_bpr_loss(anc, pos, neg):
    """loss_utils.py:7-10 (returns the SUM)."""
    pos_preds = (anc * pos).sum(-1)
    neg_preds = (anc * neg).sum(-1)
    return torch.sum(F.softplus(neg_preds - pos_preds))


def cal_infonce_loss(e1, e2, all2, temp=1.0):
    """loss_utils.py:30-39 (eps inside the sqrt, no max-subtraction, SUM)."""
    n1 = e1 / torch.sqrt(1e-8 + e1.square().sum(-1, keepdim=True))
    n2 = e2 / torch.sqrt(1e-8 + e2.square().sum(-1, keepdim=True))
    na = all2 / torch.sqrt(1e-8 + all2.square().sum(-1, keepdim=True))
    nume = -(n1 * n2 / temp).sum(-1)
    deno = torch.log(torch.sum(torch.exp(n1 @ na.T / temp), dim=-1))
    return (nume + deno).sum()


def reg_params(params):
    """loss_utils.py:20-24."""
    reg = 0
    for w in params:
        reg = reg + w.norm(2).square()
    return reg


def lightgcl_cl_terms(g_u, e_u, g_i, e_i, ancs, poss, temp):
    """LightGCL's bespoke contrastive term (lightgcl.py:114-118), un-normalized."""
    neg = torch.log(torch.exp(g_u[ancs] @ e_u.T / temp).sum(1) + 1e-8).mean()
    neg = neg + torch.log(torch.exp(g_i[poss] @ e_i.T / temp).sum(1) + 1e-8).mean()
    pos = torch.clamp((g_u[ancs] * e_u[ancs]).sum(1) / temp, -5.0, 5.0).mean() + \
        torch.clamp((g_i[poss] * e_i[poss]).sum(1) / temp, -5.0, 5.0).mean()
    return -pos + neg


def lightgcl_bpr(anc, pos, neg):
    """lightgcl.py:106-108 (-log sigmoid, MEAN)."""
    ps = (anc * pos).sum(-1)
    ns = (anc * neg).sum(-1)
    return -(ps - ns).sigmoid().log().mean()


def lightgcl_spmm(sp_adj, emb):
    """lightgcl.py:58-65: gather / scale / index_add_ over the coalesced COO."""
    sp_adj = sp_adj.coalesce()
    rows, cols = sp_adj.indices()[0], sp_adj.indices()[1]
    segs = emb[cols] * sp_adj.values().unsqueeze(1)
    out = torch.zeros((sp_adj.shape[0], emb.shape[1]))
    out.index_add_(0, rows, segs)
    return out


# --------------------------------------------------------------------------
# model-level steps (cal_loss of the four target models)
# --------------------------------------------------------------------------
def lightgcn_cal_loss(adj, ue, ie, batch, layer_num, keep_rate, reg_weight, mask_draw=None):
    """lightgcn.py:45-56."""
    u, i = lightgcn_forward(adj, ue, ie, layer_num, keep_rate, mask_draw)
    ancs, poss, negs = batch
    bpr = cal_bpr_loss(u[ancs], i[poss], i[negs]) / ancs.shape[0]
    reg = reg_weight * reg_params([ue, ie])
    return bpr + reg, {'bpr_loss': bpr, 'reg_loss': reg}


def sgl_cal_loss(adj, ue, ie, batch, layer_num, keep_rate, reg_weight, cl_weight, temp,
                 mask_draws=(None, None)):
    """sgl.py:45-65 (edge_drop augmentation)."""
    u1, i1 = lightgcn_forward(adj, ue, ie, layer_num, keep_rate, mask_draws[0])
    u2, i2 = lightgcn_forward(adj, ue, ie, layer_num, keep_rate, mask_draws[1])
    u3, i3 = lightgcn_forward(adj, ue, ie, layer_num, 1.0)
    ancs, poss, negs = batch
    bpr = cal_bpr_loss(u3[ancs], i3[poss], i3[negs]) / ancs.shape[0]
    cl = cal_infonce_loss(u1[ancs], u2[ancs], u2, temp) + \
        cal_infonce_loss(i1[poss], i2[poss], i2, temp) + \
        cal_infonce_loss(i1[negs], i2[negs], i2, temp)
    cl = cl / ancs.shape[0]
    reg = reg_weight * reg_params([ue, ie])
    cl = cl * cl_weight
    return bpr + reg + cl, {'bpr_loss': bpr, 'reg_loss': reg, 'cl_loss': cl}


def simgcl_cal_loss(adj, ue, ie, batch, layer_num, reg_weight, cl_weight, temp, eps,
                    noise_draws=(None, None)):
    """simgcl.py:39-55.  noise_draws = (view-1 draws [L], view-2 draws [L])."""
    nd1 = noise_draws[0] if noise_draws[0] is not None else [None] * layer_num
    nd2 = noise_draws[1] if noise_draws[1] is not None else [None] * layer_num
    u1, i1 = lightgcn_forward(adj, ue, ie, layer_num, noise_draws=nd1, eps=eps)
    u2, i2 = lightgcn_forward(adj, ue, ie, layer_num, noise_draws=nd2, eps=eps)
    u3, i3 = lightgcn_forward(adj, ue, ie, layer_num, 1.0)
    ancs, poss, negs = batch
    bpr = cal_bpr_loss(u3[ancs], i3[poss], i3[negs]) / ancs.shape[0]
    cl = cal_infonce_loss(u1[ancs], u2[ancs], u2, temp) + cal_infonce_loss(i1[poss], i2[poss], i2, temp)
    cl = cl / ancs.shape[0]
    reg = reg_weight * reg_params([ue, ie])
    cl = cl * cl_weight
    return bpr + reg + cl, {'bpr_loss': bpr, 'reg_loss': reg, 'cl_loss': cl}


def lightgcl_forward(adj_ui, ue, ie, ut, vt, u_mul_s, v_mul_s, layer_num):
    """lightgcl.py:73-97 with dropout p=0 (lightgcl.yml:28)."""
    e_u, e_i = [ue], [ie]
    g_u, g_i = [ue], [ie]
    adj_t = adj_ui.transpose(0, 1)
    for _ in range(layer_num):
        z_u = lightgcl_spmm(adj_ui, e_i[-1])
        z_i = lightgcl_spmm(adj_t, e_u[-1])
        g_u.append(u_mul_s @ (vt @ e_i[-1]))
        g_i.append(v_mul_s @ (ut @ e_u[-1]))
        e_u.append(z_u)
        e_i.append(z_i)
    return sum(e_u), sum(e_i), sum(g_u), sum(g_i)


def lightgcl_cal_loss(adj_ui, ue, ie, extra_params, svd, batch, layer_num, reg_weight, cl_weight, temp):
    """lightgcl.py:99-125.  `extra_params` = the unused-but-regularised Ws."""
    ut, vt, u_mul_s, v_mul_s = svd
    e_u, e_i, g_u, g_i = lightgcl_forward(adj_ui, ue, ie, ut, vt, u_mul_s, v_mul_s, layer_num)
    ancs, poss, negs = batch
    bpr = lightgcl_bpr(e_u[ancs], e_i[poss], e_i[negs])
    cl = cl_weight * lightgcl_cl_terms(g_u, e_u, g_i, e_i, ancs, poss, temp)
    reg = reg_params([ue, ie] + list(extra_params)) * reg_weight
    return bpr + cl + reg, {'bpr_loss': bpr, 'reg_loss': reg, 'cl_loss': cl}


# --------------------------------------------------------------------------
# evaluation helper (models/base_model.py:35-36, lightgcn.py:58-66)
# --------------------------------------------------------------------------
def full_predict(user_embeds, item_embeds, users, train_mask):
    preds = user_embeds[users] @ item_embeds.T
    return preds * (1 - train_mask) - 1e8 * train_mask


# --------------------------------------------------------------------------
# fp64 cross-check of the SpMM (BASELINE.md §2 "numeric sanity")
# --------------------------------------------------------------------------
def spmm_fp64(idx, vals, n_rows, x):
    a = sp.csr_matrix((vals.astype(np.float64), (idx[0], idx[1])), shape=(n_rows, x.shape[0]))
    return a @ x.astype(np.float64)
