"""Loss library of the hot path, same names / signatures / return conventions as the
reference's models/loss_utils.py (cal_bpr_loss :7-10, reg_pick_embeds :13-17, reg_params
:20-24, cal_infonce_loss :30-39), backed by the fused HIP kernels of sslrec_amd.ops.

`cal_*_gathered` are the table-level forms the in-tree models use: they take the full
embedding tables plus the batch indices, so the [B, d] gathers of lightgcn.py:49-51 /
simgcl.py:32-37 are never materialized.  The other eight functions of the upstream file
belong to models outside this path's scope and are not provided.
"""
from .. import ops


def cal_bpr_loss(anc_embeds, pos_embeds, neg_embeds, divisor=1.0):
    """sum_b softplus(<a,n> - <a,p>)  (the caller divides by the batch size -- or passes it as `divisor`, which folds
    the division and its backward into the kernels)."""
    return ops.bpr_loss(anc_embeds, pos_embeds, neg_embeds, variant=0, divisor=divisor)


def cal_bpr_loss_gathered(user_embeds, item_embeds, ancs, poss, negs, divisor=1.0):
    return ops.bpr_loss_gathered(user_embeds, item_embeds, ancs, poss, negs, variant=0, divisor=divisor)


def cal_bpr_loss_stacked(stacked_embeds, user_num, ancs, poss, negs, divisor=1.0, add=None):
    """same loss on the stacked [users; items] table that the propagation returns (no slicing).  add (a 0-d tensor, e.g. the
    regularizer term): returns (bpr + add, bpr) -- the sum comes out of the same launch"""
    return ops.bpr_loss_stacked(stacked_embeds, user_num, ancs, poss, negs, variant=0, divisor=divisor, add=add)


def cal_infonce_loss(embeds1, embeds2, all_embeds2, temp=1.0, precision=None):
    """sum_b [ -<e1^,e2^>/temp + log sum_j exp(<e1^, all^_j>/temp) ] with x^ = x/sqrt(1e-8+|x|^2).
    `precision` (not in the reference): arithmetic of the products, see ops.infonce_loss."""
    return ops.infonce_loss(embeds1, embeds2, all_embeds2, temp, variant=0, precision=precision)


def cal_infonce_loss_gathered(table1, table2, idx, temp=1.0, precision=None):
    """cal_infonce_loss(table1[idx], table2[idx], table2, temp) without the gathers."""
    return ops.infonce_loss_gathered(table1, table2, idx, temp, variant=0, precision=precision)


def cal_infonce_loss_two_sided(stacked1, stacked2, user_num, user_idx, item_idx, temp=1.0, precision=None):
    """cal_infonce_loss(U1[user_idx], U2[user_idx], U2, temp) + cal_infonce_loss(I1[item_idx], I2[item_idx], I2, temp) on the stacked
    [users; items] tables of two views: both terms of simgcl.py:49 / sgl.py:57-59 as one autograd node (no slicing of the tables)"""
    return ops.infonce_loss_two_sided(stacked1, stacked2, user_num, user_idx, item_idx, temp, variant=0, precision=precision)


def reg_pick_embeds(embeds_list):
    reg_loss = 0
    for embeds in embeds_list:
        reg_loss += embeds.square().sum()
    return reg_loss


def reg_params(model, weight=1.0):
    """weight * sum over parameters of ||W||_2^2: one fused sum-of-squares kernel per parameter (the reference
    runs `norm` + `square` and their autograd per parameter, then multiplies by reg_weight: lightgcn.py:53)"""
    reg_loss = 0
    for W in model.parameters():
        reg_loss += ops.sum_squares(W, weight)
    return reg_loss
