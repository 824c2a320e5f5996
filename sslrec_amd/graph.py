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

MAX_WAVES = 256 * 16 * 4   # streams: 256 CUs x 16 resident waves x 4 lane groups (d=64)
MIN_STREAM = 64        # do not make streams shorter than this many entries
SEG_MAX = None         # chunk cap for long rows; None = half the mean stream length (>= 64)


def _csr_arrays(rows, cols, vals, n_rows):
    """Sort COO by (row, col) (stable, duplicates kept) -> rowptr, col, val, perm."""
    order = np.lexsort((cols, rows)).astype(np.int64)
    r = rows[order]
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(r, minlength=n_rows))
    return rowptr, cols[order].astype(np.int32), vals[order].astype(np.float32), order


def _streams(rowptr, n_waves, chunk_cap, row_class=None):
    """Cut rows into segments of at most `chunk_cap` entries and deal the segments to `n_waves`
    work streams of nearly equal length (longest first, boustrophedon order).

    Returns (seg_of_pos, w_start, w_len, r_ptr, r_len, r_dst, src_index, long_row, long_ptr, n_slots):
    `src_index[e]` is the CSR position of stream entry e."""
    n_rows = rowptr.size - 1
    lens = np.diff(rowptr)
    nchunk = np.maximum(1, -(-lens // chunk_cap))           # ceil, >= 1 (empty rows still get a segment)
    is_long = nchunk > 1
    short_rows = np.nonzero(~is_long)[0]
    seg_dst = [short_rows.astype(np.int64)]
    seg_start = [rowptr[short_rows]]
    seg_len = [lens[short_rows]]
    seg_row = [short_rows]
    long_rows = np.nonzero(is_long)[0]
    long_ptr = np.zeros(long_rows.size + 1, dtype=np.int64)
    n_slots = 0
    if long_rows.size:
        nck = nchunk[long_rows]
        long_ptr[1:] = np.cumsum(nck)
        n_slots = int(long_ptr[-1])
        owner = np.repeat(np.arange(long_rows.size), nck)            # long-row index of every slot
        k = np.arange(n_slots) - long_ptr[owner]                      # chunk number inside its row
        L = lens[long_rows][owner]
        nc = nck[owner]
        base, rem = L // nc, L % nc                                    # balanced chunk sizes
        seg_dst.append(~np.arange(n_slots, dtype=np.int64))           # ~slot  (< 0)
        seg_start.append(rowptr[long_rows][owner] + k * base + np.minimum(k, rem))
        seg_len.append(base + (k < rem))
        seg_row.append(long_rows[owner])
    seg_dst = np.concatenate(seg_dst)
    seg_start = np.concatenate(seg_start)
    seg_len = np.concatenate(seg_len)
    seg_row = np.concatenate(seg_row)
    n_seg = seg_len.size
    n_waves = int(max(1, min(n_waves, n_seg)))
    # deal: longest first, snake over the streams (balances the sums without a heap)
    order = np.argsort(-seg_len, kind='stable')
    i = np.arange(n_seg)
    rnd, pos = i // n_waves, i % n_waves
    wave_of_rank = np.where(rnd % 2 == 0, pos, n_waves - 1 - pos)
    if row_class is not None:
        # streams w with (w // 4) % 8 < 4 live on XCDs 0-3 (workgroup b = w//4 runs on XCD b % 8):
        # give them class 0, the others class 1, so each XCD's L2 sees one embedding table
        cls = np.asarray(row_class)[seg_row[order]]
        xcd_half = ((np.arange(n_waves) // 4) % 8 >= 4).astype(np.int64)
        wave_of_rank = np.empty(n_seg, dtype=np.int64)
        for c in (0, 1):
            members = np.nonzero(cls == c)[0]                         # ranks of this class, longest first
            targets = np.nonzero(xcd_half == c)[0]
            if targets.size == 0:
                targets = np.arange(n_waves)
            j = np.arange(members.size)
            r2, p2 = j // targets.size, j % targets.size
            wave_of_rank[members] = targets[np.where(r2 % 2 == 0, p2, targets.size - 1 - p2)]
    # stream layout: segments grouped by wave, inside a wave in dealing order (longest first)
    by_wave = np.lexsort((i, wave_of_rank))
    seg_sorted = order[by_wave]
    r_len = seg_len[seg_sorted]
    r_dst = seg_dst[seg_sorted]
    counts = np.bincount(wave_of_rank, minlength=n_waves)
    r_ptr = np.zeros(n_waves + 1, dtype=np.int64)
    r_ptr[1:] = np.cumsum(counts)
    w_len = np.bincount(wave_of_rank, weights=seg_len[order].astype(np.float64), minlength=n_waves).astype(np.int64)
    w_start = np.zeros(n_waves, dtype=np.int64)
    w_start[1:] = np.cumsum(w_len)[:-1]
    # gather index: stream entry e <- CSR position
    total = int(r_len.sum())
    seg_off = np.zeros(r_len.size, dtype=np.int64)
    seg_off[1:] = np.cumsum(r_len)[:-1]
    src_index = np.repeat(seg_start[seg_sorted] - seg_off, r_len) + np.arange(total)
    return (w_start.astype(np.int32), w_len.astype(np.int32), r_ptr.astype(np.int32), r_len.astype(np.int32),
            r_dst.astype(np.int32), src_index, long_rows.astype(np.int32), long_ptr.astype(np.int32), n_slots)


class CsrPlan:
    """Device-resident streamed CSR of one sparse matrix (n_rows x n_cols); layout documented at
    `sslrec_csr_t` in include/sslrec_hip.h."""

    def __init__(self, rows, cols, vals, n_rows, n_cols, device, seg_max=None, share_from=None,
                 col_relabel=None, row_class=None, n_waves=None):
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
        self.csr_col_host, self.csr_val_host = col, val
        self.perm_host = perm
        self.row_class = row_class
        self._sweep = {}
        self._share_from = share_from
        same = (share_from is not None and share_from.n_rows == self.n_rows and share_from.n_cols == self.n_cols
                and np.array_equal(share_from.rowptr_host, rowptr) and np.array_equal(share_from.csr_col_host, col)
                and np.array_equal(share_from.csr_val_host, val))
        if same:                                                        # symmetric matrix: reuse device arrays
            for k in ('col', 'val', 'w_start', 'w_len', 'r_ptr', 'r_len', 'r_dst', 'long_row', 'long_ptr', 'n_slots',
                      'n_waves', 'n_rseg', 'n_long', 'src_index_host'):
                setattr(self, k, getattr(share_from, k))
            self.shared = True
        else:
            if n_waves is None:
                n_waves = min(MAX_WAVES, max(1, self.nnz // MIN_STREAM))
            if seg_max is None:
                seg_max = max(64, -(-self.nnz // max(n_waves, 1)) // 2)
            (w_start, w_len, r_ptr, r_len, r_dst, src_index, long_row, long_ptr, n_slots) = \
                _streams(rowptr, n_waves, int(seg_max), row_class)
            dev = self.device
            self.src_index_host = src_index                                 # stream entry -> CSR position
            self.col = torch.from_numpy(col[src_index]).to(dev)
            self.val = torch.from_numpy(val[src_index]).to(dev)
            self.w_start = torch.from_numpy(w_start).to(dev)
            self.w_len = torch.from_numpy(w_len).to(dev)
            self.r_ptr = torch.from_numpy(r_ptr).to(dev)
            self.r_len = torch.from_numpy(r_len).to(dev)
            self.r_dst = torch.from_numpy(r_dst).to(dev)
            self.long_row = torch.from_numpy(long_row).to(dev)
            self.long_ptr = torch.from_numpy(long_ptr).to(dev)
            self.n_slots, self.n_waves = int(n_slots), int(w_start.size)
            self.n_rseg, self.n_long = int(r_len.size), int(long_row.size)
            self.shared = False
        # stream entry -> original COO entry (for EdgeDrop masks / re-valued views)
        self.edge_map = torch.from_numpy(perm[self.src_index_host].astype(np.int32)).to(self.device)
        self._struct = None
        self._partial = {}

    def sweep(self, d):
        """SweepPlan for embedding size d, or None when the matrix does not fit the on-chip
        accumulator layout.  The sweep kernel is opt-in (SSLREC_SPMM_MODE=sweep): on MI355X it is
        LDS-bound and measured slower than the lane-group stream kernel."""
        import os
        if os.environ.get('SSLREC_SPMM_MODE', 'stream') != 'sweep':
            return None
        if d not in self._sweep:
            if self.shared and self._share_from is not None:
                sp = self._share_from.sweep(d)            # symmetric matrix: same arrays, own edge map
                if sp is not None:
                    sp = _SweepAlias(sp, self)
            else:
                sp = SweepPlan(self, d, row_class=self.row_class)
                if sp.ok:
                    sp.edge_map = torch.from_numpy(self.perm_host[sp.src_index_host].astype(np.int32)).to(self.device)
                else:
                    sp = None
            self._sweep[d] = sp
        return self._sweep[d]

    # -- C ABI view ------------------------------------------------------------------
    def c_struct(self):
        if self._struct is None:
            s = _lib.CsrStruct()
            s.n_rows, s.n_cols, s.nnz = self.n_rows, self.n_cols, self.nnz
            s.col, s.val = self.col.data_ptr(), self.val.data_ptr()
            s.n_waves = self.n_waves
            s.w_start, s.w_len, s.r_ptr = self.w_start.data_ptr(), self.w_len.data_ptr(), self.r_ptr.data_ptr()
            s.n_rseg = self.n_rseg
            s.r_len, s.r_dst = self.r_len.data_ptr(), self.r_dst.data_ptr()
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

    def algorithmic_bytes(self, d, acc=False, write_y=True):
        """Compulsory HBM traffic of one SpMM (SURVEY.md §8d formula with the stream metadata in
        place of rowptr): entries*8 + row segments*8 + streams*16 + X read once + Y written once
        (+ one read and one write of the fused accumulator)."""
        b = self.nnz * 8 + self.n_rseg * 8 + self.n_waves * 16 + self.n_cols * d * 4
        if write_y:
            b += self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


SWEEP_WPW = 16            # waves per workgroup of the sweep kernel (one workgroup per CU)
SWEEP_SPG = 10            # accumulator rows per lane group (16 x (256/d) x 10 rows x d x 4 B = 160 KiB LDS)
SWEEP_COL_BITS = 27
SWEEP_UNUSED = -2 ** 31
N_CU = 256
SWEEP_BLOCK_BYTES = 2 << 20     # column-block size of the sweep: 2 MiB of X rows (half an XCD L2)


class SweepPlan:
    """LDS-accumulator / column-block-sweep layout of a matrix for one embedding size d
    (`sslrec_sweep_t` in include/sslrec_hip.h).  Built lazily from a CsrPlan's host CSR."""

    def __init__(self, base, d, block_bytes=None, row_class=None):
        import os
        self.ok = False
        self.d = int(d)
        if d not in (32, 64, 128, 256):
            return
        n_rows, n_cols, nnz = base.n_rows, base.n_cols, base.nnz
        if n_cols >= (1 << SWEEP_COL_BITS) or n_cols * d * 4 >= (1 << 32) or nnz == 0:
            return
        lpg = d // 4
        gpw = 64 // lpg
        groups_per_wg = SWEEP_WPW * gpw
        rowptr = base.rowptr_host
        lens = np.diff(rowptr)
        n_groups_max = N_CU * groups_per_wg
        cap_rows = n_groups_max * SWEEP_SPG
        # chunk long rows so that the lane groups can be balanced; grow the cap until everything fits
        chunk = max(64, -(-nnz // n_groups_max) // 2)
        while True:
            nchunk = np.maximum(1, -(-lens // chunk))
            if int(nchunk.sum()) <= cap_rows or chunk > (1 << 30):
                break
            chunk *= 2
        n_pseudo = int(nchunk.sum())
        if n_pseudo > cap_rows:
            return                                    # does not fit on chip -> stream kernel
        is_long = nchunk > 1
        long_rows = np.nonzero(is_long)[0]
        long_ptr = np.zeros(long_rows.size + 1, dtype=np.int64)
        long_ptr[1:] = np.cumsum(nchunk[long_rows])
        n_slots = int(long_ptr[-1])
        # pseudo rows: (start, len, dst)
        owner = np.repeat(np.arange(n_rows), nchunk)
        k = np.arange(n_pseudo) - np.repeat(np.cumsum(nchunk) - nchunk, nchunk)
        L, nc = lens[owner], nchunk[owner]
        base_len, rem = L // nc, L % nc
        p_len = base_len + (k < rem)
        p_start = rowptr[owner] + k * base_len + np.minimum(k, rem)
        p_dst = owner.astype(np.int64).copy()
        long_index = np.full(n_rows, -1, dtype=np.int64)
        long_index[long_rows] = np.arange(long_rows.size)
        lm = is_long[owner]
        p_dst[lm] = ~(long_ptr[long_index[owner[lm]]] + k[lm])
        # deal pseudo rows to lane groups: longest first, snake order
        n_wg = int(min(N_CU, max(1, -(-n_pseudo // groups_per_wg))))
        n_groups = n_wg * groups_per_wg
        order = np.argsort(-p_len, kind='stable')
        i = np.arange(n_pseudo)
        rnd, pos = i // n_groups, i % n_groups
        group_of_rank = np.where(rnd % 2 == 0, pos, n_groups - 1 - pos)
        if row_class is not None:
            # workgroup b runs on XCD b % 8: class 0 on XCDs 0-3, class 1 on XCDs 4-7 (speed only)
            cls = np.asarray(row_class)[owner[order]]
            wg_of_group = np.arange(n_groups) // groups_per_wg
            half = (wg_of_group % 8 >= 4).astype(np.int64) if n_wg >= 8 else (wg_of_group % 2)
            group_of_rank = np.empty(n_pseudo, dtype=np.int64)
            slot_round = np.empty(n_pseudo, dtype=np.int64)
            ok = True
            for c in (0, 1):
                members = np.nonzero(cls == c)[0]
                targets = np.nonzero(half == c)[0]
                if targets.size == 0 or members.size > targets.size * SWEEP_SPG:
                    ok = False
                    break
                j = np.arange(members.size)
                r2, p2 = j // targets.size, j % targets.size
                group_of_rank[members] = targets[np.where(r2 % 2 == 0, p2, targets.size - 1 - p2)]
                slot_round[members] = r2
            if ok:
                rnd = slot_round
            else:
                group_of_rank = np.where(rnd % 2 == 0, pos, n_groups - 1 - pos)
        slot_of_rank = rnd                               # the r-th row dealt to a group sits in slot r
        assert int(slot_of_rank.max()) < SWEEP_SPG
        group_p = np.empty(n_pseudo, dtype=np.int64)
        slot_p = np.empty(n_pseudo, dtype=np.int64)
        group_p[order] = group_of_rank
        slot_p[order] = slot_of_rank
        # entries: gather per pseudo row, then sort by (group, column block, slot, column)
        if block_bytes is None:
            block_bytes = int(os.environ.get('SSLREC_SWEEP_BLOCK_BYTES', SWEEP_BLOCK_BYTES))
        block_rows = max(1, block_bytes // (d * 4))
        off = np.zeros(n_pseudo, dtype=np.int64)
        off[1:] = np.cumsum(p_len)[:-1]
        src = np.repeat(p_start - off, p_len) + np.arange(nnz)       # CSR position of every (pseudo-row ordered) entry
        e_group = np.repeat(group_p, p_len)
        e_slot = np.repeat(slot_p, p_len)
        e_col = base.csr_col_host[src].astype(np.int64)
        key_order = np.lexsort((e_col, e_slot, e_col // block_rows, e_group))
        src = src[key_order]
        e_group, e_slot, e_col = e_group[key_order], e_slot[key_order], e_col[key_order]
        s_len = np.bincount(e_group, minlength=n_groups).astype(np.int64)
        s_start = np.zeros(n_groups, dtype=np.int64)
        s_start[1:] = np.cumsum(s_len)[:-1]
        g_dst = np.full(n_groups * SWEEP_SPG, SWEEP_UNUSED, dtype=np.int64)
        g_dst[group_p * SWEEP_SPG + slot_p] = p_dst
        dev = base.device
        self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, nnz
        self.n_wg, self.n_groups, self.n_slots, self.n_long = n_wg, n_groups, n_slots, int(long_rows.size)
        self.block_rows, self.chunk = block_rows, int(chunk)
        self.src_index_host = src
        self.cs = torch.from_numpy((e_col | (e_slot << SWEEP_COL_BITS)).astype(np.int32)).to(dev)
        self.val = torch.from_numpy(base.csr_val_host[src]).to(dev)
        self.s_start = torch.from_numpy(s_start.astype(np.int32)).to(dev)
        self.s_len = torch.from_numpy(s_len.astype(np.int32)).to(dev)
        self.g_dst = torch.from_numpy(g_dst.astype(np.int32)).to(dev)
        self.long_row = torch.from_numpy(long_rows.astype(np.int32)).to(dev)
        self.long_ptr = torch.from_numpy(long_ptr.astype(np.int32)).to(dev)
        self.device = dev
        self._struct = None
        self._partial = None
        self.ok = True

    def c_struct(self):
        if self._struct is None:
            s = _lib.SweepStruct()
            s.n_rows, s.n_cols, s.nnz, s.d = self.n_rows, self.n_cols, self.nnz, self.d
            s.n_wg, s.n_groups = self.n_wg, self.n_groups
            s.s_start, s.s_len = self.s_start.data_ptr(), self.s_len.data_ptr()
            s.cs, s.val, s.g_dst = self.cs.data_ptr(), self.val.data_ptr(), self.g_dst.data_ptr()
            s.n_long, s.long_row, s.long_ptr = self.n_long, self.long_row.data_ptr(), self.long_ptr.data_ptr()
            s.n_slots = self.n_slots
            self._struct = s
        return self._struct

    def partial_ws(self):
        if self.n_slots == 0:
            return None
        if self._partial is None:
            self._partial = torch.empty(self.n_slots * self.d, dtype=torch.float32, device=self.device)
        return self._partial

    def algorithmic_bytes(self, d=None, acc=False, write_y=True):
        """compulsory HBM bytes of one launch: entries*8 + stream metadata + X once + Y once"""
        d = self.d
        b = self.nnz * 8 + self.n_groups * (8 + 4 * SWEEP_SPG) + self.n_cols * d * 4
        if write_y:
            b += self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


class _SweepAlias:
    """A symmetric matrix's backward SweepPlan: the forward plan's device arrays with the edge map
    of the transposed COO entries."""

    def __init__(self, sp, plan):
        self.__dict__.update(sp.__dict__)
        self._sp = sp
        self.edge_map = torch.from_numpy(plan.perm_host[sp.src_index_host].astype(np.int32)).to(plan.device)

    def c_struct(self):
        return self._sp.c_struct()

    def partial_ws(self):
        return self._sp.partial_ws()

    def algorithmic_bytes(self, d=None, acc=False, write_y=True):
        return self._sp.algorithmic_bytes(d, acc, write_y)


class PropGraph:
    """Forward + backward plans of one adjacency, and the EdgeDrop machinery."""

    def __init__(self, rows, cols, vals, shape, device, seg_max=SEG_MAX, bipartite_split=None):
        """`bipartite_split` = number of users U when the matrix is the (U+I)^2 bipartite adjacency:
        rows < U only touch columns >= U and vice versa, which the work-list order exploits to give
        each XCD's L2 one embedding table instead of two (see _xcd_class_order)."""
        n_rows, n_cols = int(shape[0]), int(shape[1])
        self.shape = (n_rows, n_cols)
        self.device = torch.device(device)
        cls_f = cls_b = None
        if bipartite_split is not None:
            cls_f = (np.arange(n_rows) >= int(bipartite_split)).astype(np.int8)
            cls_b = (np.arange(n_cols) >= int(bipartite_split)).astype(np.int8)
        self.fwd = CsrPlan(rows, cols, vals, n_rows, n_cols, device, seg_max, row_class=cls_f)
        self.bwd = CsrPlan(cols, rows, vals, n_cols, n_rows, device, seg_max, share_from=self.fwd, row_class=cls_b)
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

    def compact(self, which, d=None):
        """override arrays for plan `which` ('fwd' or 'bwd'): ('sweep', cs, val, s_len) when the
        sweep layout is in use for embedding size d, else ('stream', col, val, r_len, w_len)."""
        plan = getattr(self.graph, which)
        sp = plan.sweep(d) if d is not None else None
        key = (which, 'sweep', d) if sp is not None else (which, 'stream')
        if key not in self._compact:
            dev = plan.device
            lib = _lib.load()
            stream = torch.cuda.current_stream().cuda_stream
            if sp is not None:
                cs = torch.empty(max(plan.nnz, 1), dtype=torch.int32, device=dev)
                val = torch.empty(max(plan.nnz, 1), dtype=torch.float32, device=dev)
                s_len = torch.empty(max(sp.n_groups, 1), dtype=torch.int32, device=dev)
                rc = lib.sslrec_sweep_compact(C.byref(sp.c_struct()), sp.edge_map.data_ptr(), self.keep.data_ptr(),
                                              self.scale, cs.data_ptr(), val.data_ptr(), s_len.data_ptr(), stream)
                _lib.check(rc, 'sslrec_sweep_compact')
                self._compact[key] = ('sweep', cs, val, s_len)
            else:
                col = torch.empty(max(plan.nnz, 1), dtype=torch.int32, device=dev)
                val = torch.empty(max(plan.nnz, 1), dtype=torch.float32, device=dev)
                r_len = torch.empty(max(plan.n_rseg, 1), dtype=torch.int32, device=dev)
                w_len = torch.empty(max(plan.n_waves, 1), dtype=torch.int32, device=dev)
                rc = lib.sslrec_edge_drop_compact(C.byref(plan.c_struct()), plan.edge_map.data_ptr(),
                                                  self.keep.data_ptr(), self.scale, col.data_ptr(), val.data_ptr(),
                                                  r_len.data_ptr(), w_len.data_ptr(), stream)
                _lib.check(rc, 'sslrec_edge_drop_compact')
                self._compact[key] = ('stream', col, val, r_len, w_len)
        return self._compact[key]

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

    def compact(self, which, d=None):
        plan = getattr(self.graph, which)
        sp = plan.sweep(d) if d is not None else None
        key = (which, 'sweep', d) if sp is not None else (which, 'stream')
        if key not in self._compact:
            if sp is not None:
                self._compact[key] = ('sweep', None, self.vals[sp.edge_map.long()].contiguous(), None)
            else:
                self._compact[key] = ('stream', None, self.vals[plan.edge_map.long()].contiguous(), None, None)
        return self._compact[key]

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
