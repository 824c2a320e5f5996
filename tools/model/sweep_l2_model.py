#!/usr/bin/env python
"""MODEL, not a measurement: the L2 behaviour of the column sweep of spmm_swept_kernel, replayed on the host from the layout
arrays.  Per XCD the gathers of its 512 waves (2 lines of 128 B per 256-byte row) and their metadata loads are ordered in TIME
with a per-wave clock calibrated on the measured drift (tools/spmm_trace.py: p5-p95 spread of the block start times inside an XCD
1.4 us at block 0, 7 us in the middle, 11 us at the last block; 3.2 us per block) and run through a 4 MiB, 16-way, 128-byte-line LRU
cache.  Purpose: rank layout / scheduling ideas by the hit rate they would give BEFORE spending GPU time on them; the baseline
row must reproduce the measured counters (plain SpMM: L2 hit rate 77 %, fabric reads 200 MB per launch).
usage: python tools/model/sweep_l2_model.py [--graph amazon-book] [--d 64] [--variants base,sync,...]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sslrec_amd.data_utils import synth
from sslrec_amd.graph import PropGraph

LINE = 128
WAYS = 16
L2_BYTES = 4 << 20
SETS = L2_BYTES // LINE // WAYS


def lru_sim(lines):
    """hits of an access sequence (int64 line ids) in a SETS x WAYS LRU cache"""
    sets = [dict() for _ in range(SETS)]
    hit = np.zeros(lines.size, dtype=bool)
    # xor-folded set index (channel / set hashing of a real L2 is similar in effect: it spreads strided streams)
    idx = (lines ^ (lines >> 11) ^ (lines >> 22)) % SETS
    tick = 0
    for k in range(lines.size):
        s = sets[idx[k]]
        ln = lines[k]
        if ln in s:
            hit[k] = True
            del s[ln]
        elif len(s) >= WAYS:
            del s[next(iter(s))]
        s[ln] = tick
        tick += 1
    return hit


def wave_clock(rng, n_waves, n_blocks, T, sigma_p, sigma_r, start_jitter):
    """start time of every block of every wave [n_waves, n_blocks + 1]"""
    eps = rng.normal(0.0, sigma_p, size=(n_waves, 1))
    eta = rng.normal(0.0, sigma_r, size=(n_waves, n_blocks))
    dur = T * np.maximum(0.3, 1.0 + eps + eta)
    t0 = rng.uniform(0.0, start_jitter, size=(n_waves, 1))
    return np.concatenate([t0, t0 + np.cumsum(dur, axis=1)], axis=1)


def xcd_accesses(lay, x, d, clock_of, hot_cols=None, row_bytes=None):
    """(time, line, is_gather) of everything XCD x reads during the sweep"""
    G = lay.G
    S, LPG = lay.steps_per_block(lay.width), 64 // G
    pack = lay.pack.numpy(); ws = lay.w_start.numpy(); wst = lay.w_steps.numpy()
    row_bytes = row_bytes or d * 4
    lpr = max(1, row_bytes // LINE)                      # lines per gathered row
    waves = [w for w in range(lay.n_blocks * 16) if (w // 16) % 8 == x]
    nb_max = int(wst[waves].max()) // S
    clk = clock_of(len(waves), nb_max)
    T, L, Gt = [], [], []
    meta_base = 1 << 40
    for wi, w in enumerate(waves):
        steps = int(wst[w])
        if steps == 0:
            continue
        s = np.arange(steps)
        b = s // S
        t = clk[wi, b] + (s % S) / S * (clk[wi, b + 1] - clk[wi, b])
        for g in range(G):
            pk = pack[ws[w] + b * 64 + g * LPG + s % S]
            real = pk != -1
            col = (pk[real].view(np.uint32) & 0xFFFFF).astype(np.int64)
            tt = t[real]
            if hot_cols is not None:                     # rows kept in LDS: no request at all
                keep = ~hot_cols[col]
                col, tt = col[keep], tt[keep]
            base = col * row_bytes // LINE
            for k in range(lpr):
                T.append(tt); L.append(base + k); Gt.append(np.ones(col.size, dtype=bool))
        nblk = steps // S                                  # metadata: one 256-byte load per array per block = 4 lines
        mt = np.repeat(clk[wi, :nblk], 4)
        ml = meta_base + (ws[w] // 64 + np.repeat(np.arange(nblk), 4)) * 4 + np.tile(np.arange(4), nblk)
        T.append(mt); L.append(ml.astype(np.int64)); Gt.append(np.zeros(ml.size, dtype=bool))
    T, L, Gt = np.concatenate(T), np.concatenate(L), np.concatenate(Gt)
    o = np.argsort(T, kind='stable')
    return T[o], L[o], Gt[o]


def run(lay, d, variant, seed=0):
    rng = np.random.default_rng(seed)
    T_blk = 3.2
    drift = dict(sigma_p=0.055, sigma_r=0.12, start_jitter=1.9)       # -> p5-p95 spread 6.7 us (block 9), 11.3 us (block 17): the measured drift
    hot = None
    if variant == 'sync':                                  # every wave on the same clock
        drift = dict(sigma_p=0.0, sigma_r=0.0, start_jitter=0.0)
    elif variant == 'half_drift':
        drift = dict(sigma_p=0.0275, sigma_r=0.06, start_jitter=1.0)
    elif variant.startswith('hot'):                        # the K most gathered rows of each table live in LDS
        K = int(variant[3:])
        pk = lay.pack.numpy()
        col = (pk[pk != -1].view(np.uint32) & 0xFFFFF).astype(np.int64)
        cnt = np.bincount(col, minlength=lay.n_cols)
        hot = np.zeros(lay.n_cols, dtype=bool)
        hot[np.argsort(-cnt)[:K]] = True
    out = {'variant': variant}
    tot_g = tot_h = tot_m_meta = 0
    for x in (0, 4):                                       # one XCD of each row class; the other three behave alike
        clock_of = lambda n, nb: wave_clock(rng, n, nb, T_blk, **drift)
        T, L, isg = xcd_accesses(lay, x, d, clock_of, hot_cols=hot)
        if variant == 'ideal_front':                       # bound: every wave exactly at the same column at the same time
            o = np.argsort(np.where(isg, L, -1), kind='stable')
            T, L, isg = T[o], L[o], isg[o]
        hit = lru_sim(L)
        g, h = int(isg.sum()), int(hit[isg].sum())
        out['xcd%d' % x] = {'gather_lines': g, 'hit_rate': round(h / max(g, 1), 4), 'miss_lines': g - h,
                            'sweep_us': round(float(T.max()), 1)}
        tot_g += 4 * g; tot_h += 4 * h; tot_m_meta += 4 * int((~isg).sum())
    out['chip'] = {'gather_lines': tot_g, 'hit_rate_gathers': round(tot_h / tot_g, 4),
                   'hit_rate_all_requests': round(tot_h / (tot_g + tot_m_meta), 4),
                   'fabric_read_MB': round(((tot_g - tot_h) + tot_m_meta) * LINE / 1e6, 1)}
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--graph', default='amazon-book')
    ap.add_argument('--d', type=int, default=64)
    ap.add_argument('--variants', default='base,sync,half_drift,ideal_front,hot64,hot256')
    args = ap.parse_args()
    trn = synth.make_dataset(args.graph)
    U, I = trn.shape
    n = U + I
    keys = np.unique(trn.row.astype(np.int64) * I + trn.col)
    u, i = keys // I, keys % I + U
    rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    vals = (1.0 / np.sqrt(deg[rows] * deg[cols])).astype(np.float32)
    g = PropGraph(rows, cols, vals, (n, n), 'cpu')
    lay = g.fwd.swept(args.d)
    for v in args.variants.split(','):
        print(json.dumps(run(lay, args.d, v)), flush=True)
