"""Seeded synthetic user-item interaction graphs shaped like the datasets the
BASELINE configs name.  The reference ships only the yelp
train pickle (`.MISSING_LARGE_BLOBS`), and the GPU box has no datasets at all,
so benchmarks and large parity tests synthesize their graphs here.

The generator is plain numpy: user degrees follow a clipped power law, items
are drawn by popularity (another power law), duplicate pairs are removed and
the set is trimmed to exactly `n_edges` interactions.  Output is a scipy COO
matrix of float64 ones, i.e. the same object `pickle.load` yields for the
reference's `train_mat.pkl` (docs/GuideDataCF.md:1-33 of the reference).
"""
import numpy as np
import scipy.sparse as sp

# (users, items, train interactions) per BASELINE.json config
SHAPES = {
    'gowalla': (25557, 19747, 295000),        # cfg 1 (train pickle missing upstream)
    'amazon-book': (52643, 91599, 2380730),   # cfg 2/3 (LightGCN-paper Amazon-Book)
    'yelp': (42712, 26822, 182357),           # cfg 4 (same shape as the real pickle)
    'tiny': (300, 220, 3000),                 # unit tests / golden fixtures
}


def powerlaw_bipartite(n_user, n_item, n_edges, seed=2023, user_exp=2.0, item_exp=0.5,
                       max_user_frac=0.05):
    rng = np.random.default_rng(seed)
    # user degrees: Pareto tail, >=1, clipped, rescaled so the sum overshoots n_edges a bit
    raw = rng.pareto(user_exp - 1.0, size=n_user) + 1.0
    cap = max(1.0, max_user_frac * n_item)
    raw = np.minimum(raw, cap)
    target = n_edges * 1.15
    deg = np.maximum(1, np.floor(raw * (target / raw.sum()))).astype(np.int64)
    deg = np.minimum(deg, int(cap))
    # item popularity: Zipf-like weights over a random permutation of item ids
    w = 1.0 / np.power(np.arange(1, n_item + 1, dtype=np.float64), item_exp)
    cdf = np.cumsum(w / w.sum())
    item_of_rank = rng.permutation(n_item)

    keys = np.empty(0, dtype=np.int64)
    rounds = 0
    while keys.size < n_edges and rounds < 64:
        users = np.repeat(np.arange(n_user, dtype=np.int64), deg)
        ranks = np.searchsorted(cdf, rng.random(users.size), side='right')
        ranks = np.minimum(ranks, n_item - 1)
        items = item_of_rank[ranks]
        keys = np.unique(np.concatenate([keys, users * n_item + items]))
        rounds += 1
    if keys.size < n_edges:
        raise RuntimeError('could not reach the requested number of interactions')
    if keys.size > n_edges:
        keys = rng.choice(keys, size=n_edges, replace=False)
    # upstream pickles hold rows in no particular order: shuffle
    keys = rng.permutation(keys)
    rows = (keys // n_item).astype(np.int32)
    cols = (keys % n_item).astype(np.int32)
    return sp.coo_matrix((np.ones(n_edges, dtype=np.float64), (rows, cols)), shape=(n_user, n_item))


def community_bipartite(n_user, n_item, n_edges, n_comm=64, p_in=0.8, seed=2023, user_exp=2.0, item_exp=0.5, max_user_frac=0.05):
    """powerlaw_bipartite with PLANTED COMMUNITIES: users and items are dealt to `n_comm` communities (ids scattered over the id
    range, like real catalogues), a user draws an item from its own community with probability p_in (by popularity inside the
    community) and from the whole catalogue otherwise.  Same degree laws as powerlaw_bipartite -- the graph the locality-aware
    plan (csrc/plan.cpp: cocluster_rows) is measured on beside the structure-free headline graph."""
    rng = np.random.default_rng(seed)
    raw = rng.pareto(user_exp - 1.0, size=n_user) + 1.0
    cap = max(1.0, max_user_frac * n_item)
    raw = np.minimum(raw, cap)
    deg = np.maximum(1, np.floor(raw * (n_edges * 1.15 / raw.sum()))).astype(np.int64)
    deg = np.minimum(deg, int(cap))
    comm_u = rng.integers(0, n_comm, n_user)
    comm_i = rng.integers(0, n_comm, n_item)
    order = np.argsort(comm_i, kind='stable')                       # items grouped by community
    start = np.searchsorted(comm_i[order], np.arange(n_comm + 1))
    w = 1.0 / np.power(np.arange(1, n_item + 1, dtype=np.float64), item_exp)
    cdf = np.cumsum(w / w.sum())
    item_of_rank = rng.permutation(n_item)
    keys = np.empty(0, dtype=np.int64)
    rounds = 0
    while keys.size < n_edges and rounds < 64:
        users = np.repeat(np.arange(n_user, dtype=np.int64), deg)
        inside = rng.random(users.size) < p_in
        glob = item_of_rank[np.minimum(np.searchsorted(cdf, rng.random(users.size), side='right'), n_item - 1)]
        cu = comm_u[users]
        size = (start[cu + 1] - start[cu]).astype(np.float64)
        # inside a community: rank ~ u^(1/(1-item_exp)) reproduces the same popularity law over the community's items
        pick = np.minimum((np.power(rng.random(users.size), 1.0 / max(1e-6, 1.0 - item_exp)) * size).astype(np.int64), np.maximum(size.astype(np.int64) - 1, 0))
        loc = order[np.minimum(start[cu] + pick, n_item - 1)]
        items = np.where(inside & (size > 0), loc, glob)
        keys = np.unique(np.concatenate([keys, users * n_item + items]))
        rounds += 1
    if keys.size < n_edges:
        raise RuntimeError('could not reach the requested number of interactions')
    if keys.size > n_edges:
        keys = rng.choice(keys, size=n_edges, replace=False)
    keys = rng.permutation(keys)
    return sp.coo_matrix((np.ones(n_edges, dtype=np.float64), ((keys // n_item).astype(np.int32), (keys % n_item).astype(np.int32))),
                         shape=(n_user, n_item))


def make_dataset(name, seed=2023):
    n_user, n_item, n_edges = SHAPES[name]
    return powerlaw_bipartite(n_user, n_item, n_edges, seed)


def split_holdout(trn_mat, frac, seed):
    """Tiny helper for fixtures: a disjoint random hold-out with the same shape
    (stands in for valid_mat / test_mat)."""
    rng = np.random.default_rng(seed)
    n_user, n_item = trn_mat.shape
    n = max(1, int(frac * trn_mat.nnz))
    existing = set((trn_mat.row.astype(np.int64) * n_item + trn_mat.col).tolist())
    picked = []
    while len(picked) < n:
        k = int(rng.integers(0, n_user * n_item))
        if k not in existing:
            existing.add(k)
            picked.append(k)
    keys = np.array(picked, dtype=np.int64)
    return sp.coo_matrix((np.ones(n), ((keys // n_item).astype(np.int32), (keys % n_item).astype(np.int32))),
                         shape=(n_user, n_item))


def cell_bipartite(n_user, n_item, n_edges, world, ru, ri, seed=2023):
    """Shard-local generation for row-sharded training (BASELINE cfg 5: a 10 M x 10 M graph nobody can build on one
    host): the global graph is DEFINED as the union of world^2 cells, cell (ru, ri) holding the interactions between
    users = ru mod world and items = ri mod world, generated from its own seed.  Rank p builds cells (p, *) for its rows
    of A and cells (*, p) for its rows of A^T -- 2/world of the graph, no communication; degrees are exchanged
    afterwards (`sharded_lightgcl_values`).  Returns (users, items) int64 global ids of the cell."""
    uc = (n_user - ru + world - 1) // world
    ic = (n_item - ri + world - 1) // world
    ec = n_edges // (world * world) + (1 if (ru * world + ri) < n_edges % (world * world) else 0)
    if uc <= 0 or ic <= 0 or ec <= 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    m = powerlaw_bipartite(uc, ic, min(ec, uc * ic // 2), seed + 7919 * (ru * world + ri))
    return m.row.astype(np.int64) * world + ru, m.col.astype(np.int64) * world + ri


def sharded_cells(n_user, n_item, n_edges, world, rank, seed=2023):
    """(users_f, items_f), (users_b, items_b): the interactions of this rank's user rows and of its item rows"""
    fu, fi, bu, bi = [], [], [], []
    for k in range(world):
        u, i = cell_bipartite(n_user, n_item, n_edges, world, rank, k, seed)
        fu.append(u); fi.append(i)
        u, i = cell_bipartite(n_user, n_item, n_edges, world, k, rank, seed)
        bu.append(u); bi.append(i)
    return (np.concatenate(fu), np.concatenate(fi)), (np.concatenate(bu), np.concatenate(bi))
