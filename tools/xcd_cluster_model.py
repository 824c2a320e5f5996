#!/usr/bin/env python
"""Host-side evidence for the locality-aware plan (csrc/plan.cpp: cocluster_rows, option "xcd_cluster"): the distinct (XCD, column)
pairs of the column-swept layout -- x 4 d bytes = the bytes the eight L2s pull through the fabric per launch when every line is
fetched once per XCD -- with rows dealt to the XCDs by load only and with the co-clustering, on four graphs:
  amazon-book-shaped headline graph (item_exp 0.5, no structure) | the same generator at item_exp 1.0 | planted communities |
  the REAL yelp interactions (tests/golden).        No GPU needed: the plan builder is host code.
usage: python tools/xcd_cluster_model.py [out.json]"""
import ctypes as C, json, os, sys, time
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_expr as R
from sslrec_amd import _lib
from sslrec_amd.data_utils import synth


def layouts(trn, d, passes_list):
    idx, vals, n = R.normalized_bipartite_coo(R.binarize_coo(trn))
    lib = _lib.load()
    rows, cols = np.ascontiguousarray(idx[0], dtype=np.int64), np.ascontiguousarray(idx[1], dtype=np.int64)
    v = np.ascontiguousarray(vals, dtype=np.float32)
    out = {'n_rows': int(n), 'nnz': int(v.size), 'table_MB': n * d * 4 / 1e6}
    for passes in passes_list:
        h = C.c_void_p()
        _lib.check(lib.sslrec_plan_build_coo(rows.ctypes.data, cols.ctypes.data, v.ctypes.data, v.size, n, n, C.byref(h)), 'build')
        _lib.check(lib.sslrec_plan_set_option(h, b'xcd_cluster', passes), 'opt')
        t0 = time.time()
        assert lib.sslrec_plan_layout(h, d, 1, 0) == 1, 'no swept layout'
        inf = _lib.PlanInfoStruct()
        _lib.check(lib.sslrec_plan_info(h, d, 1, C.byref(inf)), 'info')
        out['automatic' if passes == 17 else 'passes_%d' % passes] = {'xcd_col_pairs': int(inf.xcd_col_pairs), 'fabric_floor_MB': inf.xcd_col_pairs * d * 4 / 1e6,
                                     'times_the_table': inf.xcd_col_pairs / n, 'xcd_split': bool(inf.xcd_split), 'n_blocks': inf.n_blocks,
                                     'n_slots': inf.n_slots, 'build_s': round(time.time() - t0, 2)}
        lib.sslrec_plan_free(h)
    return out


if __name__ == '__main__':
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'yelp_lightgcn_d64_L2.npz'))
    U, I = (int(x) for x in z['shape'])
    yelp = sp.coo_matrix((np.ones(z['trn_row'].size), (z['trn_row'], z['trn_col'])), shape=(U, I))
    u, i, e = synth.SHAPES['amazon-book']
    graphs = {
        'amazon-book-shaped, item_exp 0.5 (the headline graph)': synth.make_dataset('amazon-book'),
        'amazon-book-shaped, item_exp 1.0': synth.powerlaw_bipartite(u, i, e, item_exp=1.0),
        'amazon-book-shaped, 64 planted communities, p_in 0.95': synth.community_bipartite(u, i, e, 64, 0.95),
        'real yelp interactions': yelp,
    }
    res = {}
    for name, g in graphs.items():
        res[name] = layouts(g, 64, (0, 4, 17))
        print(name, json.dumps(res[name]), flush=True)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], 'w'), indent=1)
