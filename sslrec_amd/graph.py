"""CSR work-list plans for the HIP SpMM (host side, numpy + torch for device memory).

The reference hands its models a `torch.sparse` COO adjacency whose entries are in an
arbitrary (in practice column-major) order and re-coalesces it inside every `t.spmm`
call.  Here the COO is converted ONCE into

  * a CSR of A   (forward  Y = A X)         and
  * a CSR of A^T (backward dX = A^T dY; shares the device arrays with A when A is
    symmetric, which the normalized bipartite adjacency is),

each with a work list of row segments (see include/sslrec_hip.h) and an `edge_map`
(CSR position -> original COO entry) so the reference's per-entry EdgeDrop mask
(models/aug_utils.py:28, drawn in COO order) can be applied without rebuilding anything.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

SEG_MAX = 128        # longest run of entries one wavefront walks; longer rows are chunked


def _csr_arrays(rows, cols, vals, n_rows):
    """Sort COO by (row, col) (stable, duplicates kept) -> rowptr, col, val, perm."""
    order = np.lexsort((cols, rows)).astype(np.int64)
    r = rows[order]
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(r, minlength=n_rows))
    return rowptr, cols[order].astype(np.int32), vals[order].astype(np.float32), order


def _segments(rowptr, seg_max):
    """Cut rows into segments of at most seg_max entries; returns the work list sorted by
    decreasing length plus the long-row bookkeeping."""
    n_rows = rowptr.size - 1
    lens = np.diff(rowptr)
    nchunk = np.maximum(1, -(-lens // seg_max))             # ceil, >= 1 (empty rows still get a segment)
    is_long = nchunk > 1
    # short rows (incl. empty): one segment each, written straight to the output row
    short_rows = np.nonzero(~is_long)[0]
    seg_dst = [short_rows.astype(np.int64)]
    seg_start = [rowptr[short_rows]]
    seg_len = [lens[short_rows]]
    long_rows = np.nonzero(is_long)[0]
    long_ptr = np.zeros(long_rows.size + 1, dtype=np.int64)
    if long_rows.size:
        nck = nchunk[long_rows]
        long_ptr[1:] = np.cumsum(nck)
        n_slots = int(long_ptr[-1])
        owner = np.repeat(np.arange(long_rows.size), nck)            # long-row index of every slot
        k = np.arange(n_slots) - long_ptr[owner]                      # chunk number inside its row
        L = lens[long_rows][owner]
        nc = nck[owner]
        base, rem = L // nc, L % nc                                    # balanced chunk sizes
        clen = base + (k < rem)
        cstart = rowptr[long_rows][owner] + k * base + np.minimum(k, rem)
        seg_dst.append(~np.arange(n_slots, dtype=np.int64))           # ~slot  (< 0)
        seg_start.append(cstart)
        seg_len.append(clen)
    else:
        n_slots = 0
    seg_dst = np.concatenate(seg_dst)
    seg_start = np.concatenate(seg_start)
    seg_len = np.concatenate(seg_len)
    order = np.argsort(-seg_len, kind='stable')
    return (seg_dst[order].astype(np.int32), seg_start[order].astype(np.int32), seg_len[order].astype(np.int32),
            long_rows.astype(np.int32), long_ptr.astype(np.int32), n_slots)


class CsrPlan:
    """Device-resident CSR + work list of one sparse matrix (n_rows x n_cols)."""

    def __init__(self, rows, cols, vals, n_rows, n_cols, device, seg_max=SEG_MAX, share_from=None,
                 col_relabel=None):
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float32)
        if rows.size >= 2 ** 31 - 1:
            raise ValueError('a single shard is limited to 2^31-1 entries (int32 CSR)')
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(rows.size)
        self.device = torch.device(device)
        rowptr, col, val, perm = _csr_arrays(rows, cols, vals, n_rows)
        if col_relabel is not None:      # new column ids, entry order (= summation order) unchanged
            col = np.asarray(col_relabel(col.astype(np.int64))).astype(np.int32)
        self.rowptr_host = rowptr
        self.col_host, self.val_host = col, val
        self.perm_host = perm                                           # CSR position -> COO entry
        same = (share_from is not None and share_from.n_rows == self.n_rows and share_from.n_cols == self.n_cols
                and np.array_equal(share_from.rowptr_host, rowptr) and np.array_equal(share_from.col_host, col)
                and np.array_equal(share_from.val_host, val))
        if same:                                                        # symmetric matrix: reuse device arrays
            for k in ('col', 'val', 'seg_dst', 'seg_start', 'seg_len', 'long_row', 'long_ptr', 'n_slots',
                      'n_seg', 'n_long'):
                setattr(self, k, getattr(share_from, k))
            self.shared = True
        else:
            seg_dst, seg_start, seg_len, long_row, long_ptr, n_slots = _segments(rowptr, seg_max)
            dev = self.device
            self.col = torch.from_numpy(col).to(dev)
            self.val = torch.from_numpy(val).to(dev)
            self.seg_dst = torch.from_numpy(seg_dst).to(dev)
            self.seg_start = torch.from_numpy(seg_start).to(dev)
            self.seg_len = torch.from_numpy(seg_len).to(dev)
            self.long_row = torch.from_numpy(long_row).to(dev)
            self.long_ptr = torch.from_numpy(long_ptr).to(dev)
            self.n_slots, self.n_seg, self.n_long = int(n_slots), int(seg_dst.size), int(long_row.size)
            self.shared = False
        self.edge_map = torch.from_numpy(perm.astype(np.int32)).to(self.device)
        self._struct = None
        self._partial = {}

    # -- C ABI view ------------------------------------------------------------------
    def c_struct(self):
        if self._struct is None:
            s = _lib.CsrStruct()
            s.n_rows, s.n_cols, s.nnz = self.n_rows, self.n_cols, self.nnz
            s.col, s.val = self.col.data_ptr(), self.val.data_ptr()
            s.n_seg = self.n_seg
            s.seg_dst, s.seg_start, s.seg_len = self.seg_dst.data_ptr(), self.seg_start.data_ptr(), self.seg_len.data_ptr()
            s.n_long = self.n_long
            s.long_row, s.long_ptr = self.long_row.data_ptr(), self.long_ptr.data_ptr()
            s.n_slots = self.n_slots
            self._struct = s
        return self._struct

    def partial_ws(self, d):
        """scratch slab for the chunk partial sums of long rows (n_slots x d floats)"""
        if self.n_slots == 0:
            return None
        if d not in self._partial:
            self._partial[d] = torch.empty(self.n_slots * d, dtype=torch.float32, device=self.device)
        return self._partial[d]

    def algorithmic_bytes(self, d, masked=False, acc=False):
        """Compulsory HBM traffic of one SpMM (SURVEY.md §8d formula, with the work list
        standing in for rowptr): entries*8 + segments*12 + X read once + Y written once."""
        b = self.nnz * 8 + self.n_seg * 12 + self.n_cols * d * 4 + self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


class PropGraph:
    """Forward + backward plans of one adjacency, and the EdgeDrop machinery."""

    def __init__(self, rows, cols, vals, shape, device, seg_max=SEG_MAX):
        n_rows, n_cols = int(shape[0]), int(shape[1])
        self.shape = (n_rows, n_cols)
        self.device = torch.device(device)
        self.fwd = CsrPlan(rows, cols, vals, n_rows, n_cols, device, seg_max)
        self.bwd = CsrPlan(cols, rows, vals, n_cols, n_rows, device, seg_max, share_from=self.fwd)
        self.nnz = self.fwd.nnz

    @classmethod
    def _single(cls, rows, cols, vals, shape, device, seg_max=SEG_MAX, col_relabel=None, share_from=None):
        """forward-only graph (one plan), used for the row shards of sslrec_amd.shard"""
        g = object.__new__(cls)
        g.shape = (int(shape[0]), int(shape[1]))
        g.device = torch.device(device)
        g.fwd = CsrPlan(rows, cols, vals, g.shape[0], g.shape[1], device, seg_max,
                        share_from=share_from.fwd if share_from is not None else None, col_relabel=col_relabel)
        g.bwd = None
        g.nnz = g.fwd.nnz
        return g

    @classmethod
    def from_torch_sparse(cls, adj, device=None, seg_max=SEG_MAX):
        """`adj`: torch sparse COO tensor exactly as the reference's data handler builds it
        (uncoalesced, any entry order; data_handler_general_cf.py:70-73)."""
        idx = adj._indices().detach().cpu().numpy()
        val = adj._values().detach().cpu().numpy()
        return cls(idx[0], idx[1], val, tuple(adj.shape), device if device is not None else adj.device, seg_max)

    def transposed(self):
        """View with forward/backward swapped (LightGCL multiplies by A and by A^T)."""
        t = object.__new__(PropGraph)
        t.shape = (self.shape[1], self.shape[0])
        t.device, t.fwd, t.bwd, t.nnz = self.device, self.bwd, self.fwd, self.nnz
        return t


class DroppedView:
    """An edge-dropped view of a PropGraph: the result of EdgeDrop (aug_utils.py:18-31) without
    rebuilding the sparse tensor.  `keep` is the reference's per-COO-entry boolean mask."""

    def __init__(self, graph, keep, scale=1.0):
        self.graph = graph
        self.keep = keep.to(device=graph.device, dtype=torch.uint8).contiguous()
        if self.keep.numel() != graph.nnz:
            raise ValueError('mask length %d != number of entries %d' % (self.keep.numel(), graph.nnz))
        self.scale = float(scale)
        self.shape = graph.shape
        self._compact = {}

    def compact(self, which):
        """(col, val, seg_len) override arrays for plan `which` ('fwd' or 'bwd')."""
        if which not in self._compact:
            plan = getattr(self.graph, which)
            dev = plan.device
            col = torch.empty(max(plan.nnz, 1), dtype=torch.int32, device=dev)
            val = torch.empty(max(plan.nnz, 1), dtype=torch.float32, device=dev)
            seg_len = torch.empty(max(plan.n_seg, 1), dtype=torch.int32, device=dev)
            lib = _lib.load()
            rc = lib.sslrec_edge_drop_compact(C.byref(plan.c_struct()), plan.edge_map.data_ptr(), self.keep.data_ptr(),
                                              self.scale, col.data_ptr(), val.data_ptr(), seg_len.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, 'sslrec_edge_drop_compact')
            self._compact[which] = (col, val, seg_len)
        return self._compact[which]

    def n_kept(self):
        return int(self.keep.sum().item())


class RevaluedView:
    """A PropGraph whose entry VALUES are replaced (same pattern), e.g. after
    `F.dropout(adj.values(), p)` in LightGCL's `_sparse_dropout` (lightgcl.py:67-71).
    `vals` is given in the original COO entry order."""

    def __init__(self, graph, vals):
        self.graph = graph
        self.vals = vals.to(device=graph.device, dtype=torch.float32).contiguous()
        if self.vals.numel() != graph.nnz:
            raise ValueError('value count %d != number of entries %d' % (self.vals.numel(), graph.nnz))
        self.shape = graph.shape
        self._compact = {}

    def compact(self, which):
        if which not in self._compact:
            plan = getattr(self.graph, which)
            self._compact[which] = (None, self.vals[plan.edge_map.long()].contiguous(), None)
        return self._compact[which]

    def transposed(self):
        t = object.__new__(RevaluedView)
        t.graph, t.vals, t.shape = self.graph.transposed(), self.vals, (self.shape[1], self.shape[0])
        t._compact = {}
        return t


_GRAPH_CACHE = {}


def graph_of(adj):
    """PropGraph of a torch sparse adjacency, built once and cached on the tensor object (and
    by storage identity) -- the CSR conversion the reference repeats on every spmm call."""
    if isinstance(adj, (PropGraph, DroppedView, RevaluedView)):
        return adj
    g = getattr(adj, '_sslrec_graph', None)
    if g is not None:
        return g
    key = (adj._indices().data_ptr(), adj._values().data_ptr(), adj._nnz(), tuple(adj.shape), str(adj.device))
    g = _GRAPH_CACHE.get(key)
    if g is None:
        g = PropGraph.from_torch_sparse(adj)
        _GRAPH_CACHE[key] = g
    try:
        adj._sslrec_graph = g
    except Exception:
        pass
    return g
