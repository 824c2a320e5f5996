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

MAX_WAVES = 256 * 20   # one stream per resident wavefront: 256 CUs x 4 SIMDs x 5 waves (94 VGPRs)
MIN_STREAM = 64        # do not make streams shorter than this many entries
ROW_OVERHEAD = 6       # cost of finishing a row segment, in entry-equivalents, for the balancing heuristic
SEG_MAX = None         # chunk cap for long rows; None = half the mean stream length (>= 64)


def _csr_arrays(rows, cols, vals, n_rows):
    """Sort COO by (row, col) (stable, duplicates kept) -> rowptr, col, val, perm."""
    order = np.lexsort((cols, rows)).astype(np.int64)
    r = rows[order]
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(r, minlength=n_rows))
    return rowptr, cols[order].astype(np.int32), vals[order].astype(np.float32), order


def _deal(lens_desc, targets):
    """Assign items (lengths sorted in decreasing order) to `targets` so that the per-target sums
    are balanced: longest-processing-time-first with a heap (exact greedy) up to 2M items, a
    boustrophedon deal beyond that.  A segment costs its entries plus a fixed per-row overhead."""
    n, m = lens_desc.size, targets.size
    if n > 2_000_000:
        j = np.arange(n)
        rnd, pos = j // m, j % m
        return targets[np.where(rnd % 2 == 0, pos, m - 1 - pos)]
    import heapq
    heap = [(0, int(k)) for k in range(m)]
    out = np.empty(n, dtype=np.int64)
    cost = lens_desc.astype(np.int64) + ROW_OVERHEAD
    for j in range(n):
        load, k = heap[0]
        out[j] = k
        heapq.heapreplace(heap, (load + int(cost[j]), k))
    return targets[out]


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
    # deal the segments to the streams, longest first
    order = np.argsort(-seg_len, kind='stable')
    i = np.arange(n_seg)
    phase = np.zeros(n_seg, dtype=np.int64)
    if row_class is None:
        wave_of_rank = _deal(seg_len[order], np.arange(n_waves))
    else:
        # TEMPORAL class phases: every stream gets its share of class-0 rows (e.g. user rows, which
        # gather ITEM embeddings) and walks them FIRST, then its class-1 rows.  All streams start
        # together, so in the first half of the launch the whole chip gathers from one embedding
        # table and in the second half from the other: the live working set in every XCD's L2 is
        # one table instead of two, and -- unlike a spatial split over XCDs -- every XCD sees the
        # same mix, so nothing goes out of balance.
        cls = np.asarray(row_class)[seg_row[order]]
        wave_of_rank = np.empty(n_seg, dtype=np.int64)
        for c in (0, 1):
            members = np.nonzero(cls == c)[0]                         # ranks of this class, longest first
            wave_of_rank[members] = _deal(seg_len[order][members], np.arange(n_waves))
        phase = cls.astype(np.int64)
    # stream layout: segments grouped by wave, inside a wave in dealing order (longest first)
    by_wave = np.lexsort((i, phase, wave_of_rank))
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


def slot_elem(k, G):
    """element offset (inside its stream) of slot k: blocks of 4 loads, stored lane-group-major
    (see sslrec_csr_t in include/sslrec_hip.h)"""
    load, sub = k // G, k % G
    return (load >> 2) * (4 * G) + sub * 4 + (load & 3)


class PackedLayout:
    """Device arrays of one streamed CSR packed for ONE embedding size d (`sslrec_csr_t`)."""

    def __init__(self, plan, d, alias_of=None):
        self.d, self.G = int(d), 256 // int(d)
        self.n_rows, self.n_cols, self.nnz, self.device = plan.n_rows, plan.n_cols, plan.nnz, plan.device
        self.n_waves, self.n_rseg, self.n_long, self.n_slots = plan.n_waves, plan.n_rseg, plan.n_long, plan.n_slots
        self.r_ptr, self.r_dst, self.long_row, self.long_ptr = plan.r_ptr, plan.r_dst, plan.long_row, plan.long_ptr
        G = self.G
        if alias_of is not None:          # symmetric matrix: A^T reuses A's arrays, only the edge map differs
            for k in ('col', 'val', 'w_start', 'w_len', 'r_len', 'n_elem', 'elem_of_entry_host'):
                setattr(self, k, getattr(alias_of, k))
        else:
            seg_len = plan.r_len_entries_host.astype(np.int64)                 # entries per row segment (stream order)
            seg_loads = -(-seg_len // G)
            seg_wave = np.repeat(np.arange(plan.n_waves), np.diff(plan.r_ptr_host))
            w_loads = np.bincount(seg_wave, weights=seg_loads.astype(np.float64), minlength=plan.n_waves).astype(np.int64)
            w_elems = -(-w_loads // 4) * 4 * G                                   # whole blocks of 4 loads
            w_start = np.zeros(plan.n_waves, dtype=np.int64)
            w_start[1:] = np.cumsum(w_elems)[:-1]
            n_elem = int(w_elems.sum())
            if n_elem >= 2 ** 31 - 1:
                raise ValueError('packed layout exceeds int32 indexing')
            # slot of every real entry: (slots of earlier segments of its stream) + position in its segment
            seg_slot0 = np.cumsum(seg_loads * G) - seg_loads * G                 # running over ALL segments ...
            first_seg = plan.r_ptr_host[:-1]
            wave_slot0 = np.zeros(plan.n_waves, dtype=np.int64)
            has = np.diff(plan.r_ptr_host) > 0
            wave_slot0[has] = seg_slot0[first_seg[has]]
            seg_slot0 = seg_slot0 - wave_slot0[seg_wave]                         # ... made relative to the stream
            off = np.cumsum(seg_len) - seg_len
            e_seg = np.repeat(np.arange(seg_len.size), seg_len)
            e_slot = seg_slot0[e_seg] + (np.arange(plan.nnz) - off[e_seg])
            elem = w_start[seg_wave[e_seg]] + slot_elem(e_slot, G)
            col = np.full(max(n_elem, 1), -1, dtype=np.int32)
            val = np.zeros(max(n_elem, 1), dtype=np.float32)
            col[elem] = plan.csr_col_host[plan.src_index_host]
            val[elem] = plan.csr_val_host[plan.src_index_host]
            dev = self.device
            self.n_elem = n_elem
            self.elem_of_entry_host = elem
            self.col = torch.from_numpy(col).to(dev)
            self.val = torch.from_numpy(val).to(dev)
            self.w_start = torch.from_numpy(w_start.astype(np.int32)).to(dev)
            self.w_len = torch.from_numpy(w_loads.astype(np.int32)).to(dev)
            self.r_len = torch.from_numpy(seg_loads.astype(np.int32)).to(dev)
        emap = np.full(max(self.n_elem, 1), -1, dtype=np.int32)
        emap[self.elem_of_entry_host] = plan.perm_host[plan.src_index_host]
        self.edge_map = torch.from_numpy(emap).to(self.device)                # element -> original COO entry
        self._struct = None
        self._partial = None

    def c_struct(self):
        if self._struct is None:
            s = _lib.CsrStruct()
            s.n_rows, s.n_cols, s.nnz, s.d, s.n_elem = self.n_rows, self.n_cols, self.nnz, self.d, self.n_elem
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

    def partial_ws(self):
        """scratch slab for the chunk partial sums of long rows (n_slots x d floats)"""
        if self.n_slots == 0:
            return None
        if self._partial is None:
            self._partial = torch.empty(self.n_slots * self.d, dtype=torch.float32, device=self.device)
        return self._partial

    def algorithmic_bytes(self, d=None, acc=False, write_y=True):
        """Compulsory HBM traffic of one SpMM (SURVEY.md §8d formula with the stream metadata in
        place of rowptr): real entries*8 + row segments*8 + streams*16 + X read once + Y written
        once (+ one read and one write of the fused accumulator).  Pads are NOT counted."""
        d = self.d
        b = self.nnz * 8 + self.n_rseg * 8 + self.n_waves * 16 + self.n_cols * d * 4
        if write_y:
            b += self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


SWEPT_BLOCKS = 256          # one workgroup per CU
SWEPT_WAVES = 16            # 1024-thread workgroups
SWEPT_LDS_BYTES = 163840    # SSLREC_SWEPT_LDS_BYTES: all of a CU's LDS holds accumulators


def swept_enabled():
    import os
    return os.environ.get('SSLREC_SPMM_SWEPT', '1') != '0'


class SweptLayout:
    """Column-swept layout (`sslrec_swept_t`, kernel sslrec_amd/csrc/spmm_swept.hip) of one plan for one
    embedding size: output rows live in LDS, every lane group owns a disjoint set of them and walks its
    edges sorted by column.  Only for matrices whose OUTPUT table fits the chip's LDS; `fits()` says so."""

    @staticmethod
    def fits(n_rows, n_cols, d):
        return (n_rows * d * 4 <= 0.985 * SWEPT_BLOCKS * SWEPT_LDS_BYTES and n_cols <= (1 << 20)
                and SWEPT_LDS_BYTES // (d * 4) <= 4095)

    def __init__(self, plan, d):
        import heapq
        d = int(d)
        G = 256 // d
        nb, nw = SWEPT_BLOCKS, SWEPT_WAVES
        slot_cap = SWEPT_LDS_BYTES // (d * 4)
        n, nnz = plan.n_rows, plan.nnz
        rowptr = plan.rowptr_host
        deg = np.diff(rowptr)
        gpb = nw * G                                    # lane groups per block
        # heavy rows are cut into INTERLEAVED chunks (entry j of the row -> chunk j % n_chunks), each with its
        # own accumulator slot; the smallest cap whose slots still fit balances the lane groups best
        for factor in (0.4, 0.6, 1.0, 2.0, 4.0, 16.0, 1e9):
            chunk_cap = max(16, int(factor * nnz / (nb * gpb)))
            n_chunks = np.maximum(1, -(-deg // chunk_cap))
            if int(n_chunks.sum()) <= 0.985 * nb * slot_cap and int(n_chunks.max()) <= slot_cap // 2:
                break
        else:
            raise ValueError('output table does not fit the LDS of %d workgroups' % nb)
        # a row is accumulated by ONE workgroup: a matrix dominated by a few giant rows would serialize on their
        # blocks -- the streamed kernel spreads such rows over the whole chip instead
        if n and int(deg.max()) > max(8192, 4 * (nnz // nb)):
            raise ValueError('row of %d entries against %d per workgroup: use the streamed kernel' % (int(deg.max()), nnz // nb))
        # rows -> blocks: longest-processing-time-first on entries, at most slot_cap slots per block
        heap = [(0, b) for b in range(nb)]
        used = np.zeros(nb, dtype=np.int64)
        blk_of_row = np.empty(n, dtype=np.int64)
        deg_l, nch_l = deg.tolist(), n_chunks.tolist()
        for r in np.argsort(-deg, kind='stable').tolist():
            parked = []
            while True:
                load, b = heapq.heappop(heap)
                if used[b] + nch_l[r] <= slot_cap:
                    break
                parked.append((load, b))
            blk_of_row[r] = b
            used[b] += nch_l[r]
            if used[b] < slot_cap:
                heapq.heappush(heap, (load + deg_l[r], b))
            for it in parked:
                heapq.heappush(heap, it)
        # slots: a block's rows in row order, the chunks of a row contiguous (the flush adds them in order)
        row_order = np.lexsort((np.arange(n), blk_of_row))
        blk_sorted, nch_sorted = blk_of_row[row_order], n_chunks[row_order]
        first_of_blk = np.searchsorted(blk_sorted, np.arange(nb))
        cum = np.cumsum(nch_sorted) - nch_sorted
        slot_sorted = cum - cum[np.minimum(first_of_blk, max(n - 1, 0))][blk_sorted] if n else cum
        slot_start = np.empty(n, dtype=np.int64)
        slot_start[row_order] = slot_sorted
        # chunks -> lane groups of their block, LPT again
        v_first = np.cumsum(n_chunks) - n_chunks
        v_row = np.repeat(np.arange(n), n_chunks)
        v_chunk = np.arange(v_row.size) - v_first[v_row]
        v_len = deg[v_row] // n_chunks[v_row] + (v_chunk < deg[v_row] % n_chunks[v_row])
        v_blk = blk_of_row[v_row]
        v_grp = np.empty(v_row.size, dtype=np.int64)
        order_v = np.lexsort((-v_len, v_blk))
        bounds = np.searchsorted(v_blk[order_v], np.arange(nb + 1))
        v_len_l = v_len.tolist()
        for b in range(nb):
            ids = order_v[bounds[b]:bounds[b + 1]].tolist()
            h = [(0, g) for g in range(gpb)]
            for v in ids:
                load, g = h[0]
                v_grp[v] = g
                heapq.heapreplace(h, (load + v_len_l[v], g))
        # entries (CSR order: by row, then column) -> (lane group, slot); streams sorted by column
        e_row = np.repeat(np.arange(n), deg)
        pos_in_row = np.arange(nnz) - rowptr[e_row]
        e_chunk = pos_in_row % n_chunks[e_row]
        e_slot = slot_start[e_row] + e_chunk
        gid = blk_of_row[e_row] * gpb + v_grp[v_first[e_row] + e_chunk]
        col = plan.csr_col_host.astype(np.int64)
        o = np.lexsort((col, gid))
        gid_s = gid[o]
        g_len = np.bincount(gid_s, minlength=nb * gpb)
        s_in_g = np.arange(nnz) - (np.cumsum(g_len) - g_len)[gid_s]
        w_steps = -(-g_len.reshape(-1, G).max(1) // 4) * 4
        w_start = np.cumsum(w_steps * G) - w_steps * G
        n_elem = int((w_steps * G).sum())
        if n_elem >= 2 ** 31 - 1:
            raise ValueError('swept layout exceeds int32 indexing')
        elem = w_start[gid_s // G] + (s_in_g // 4) * (4 * G) + (gid_s % G) * 4 + (s_in_g % 4)
        pack = np.full(max(n_elem, 1), -1, dtype=np.int32)
        val = np.zeros(max(n_elem, 1), dtype=np.float32)
        pack[elem] = (col[o] | (e_slot[o] << 20)).astype(np.uint32).view(np.int32)
        val[elem] = plan.csr_val_host[o]
        dev = plan.device
        self.d, self.G, self.n_rows, self.n_cols, self.nnz = d, G, n, plan.n_cols, nnz
        self.n_elem, self.n_blocks, self.n_slots = n_elem, nb, max(1, int(used.max()))
        self.chunk_cap, self.device = chunk_cap, dev
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
        self.pack, self.val = torch.from_numpy(pack).to(dev), torch.from_numpy(val).to(dev)
        self.w_start, self.w_steps = t(w_start, np.int32), t(w_steps, np.int32)
        self.f_ptr = t(np.concatenate([first_of_blk, [n]]), np.int32)
        self.f_row, self.f_start, self.f_n = t(row_order, np.int32), t(slot_sorted, np.int32), t(nch_sorted, np.int32)
        self.n_flush = int(n)
        self.elem_host, self.csr_pos_host = elem, o      # element of the i-th (lane group, column)-sorted entry / its CSR position
        self._struct = None

    def c_struct(self):
        if self._struct is None:
            s = _lib.SweptStruct()
            s.n_rows, s.n_cols, s.nnz, s.d = self.n_rows, self.n_cols, self.nnz, self.d
            s.n_elem, s.n_blocks, s.n_slots = self.n_elem, self.n_blocks, self.n_slots
            s.pack, s.val = self.pack.data_ptr(), self.val.data_ptr()
            s.w_start, s.w_steps = self.w_start.data_ptr(), self.w_steps.data_ptr()
            s.f_ptr, s.f_row = self.f_ptr.data_ptr(), self.f_row.data_ptr()
            s.f_start, s.f_n = self.f_start.data_ptr(), self.f_n.data_ptr()
            self._struct = s
        return self._struct

    def algorithmic_bytes(self, d=None, acc=False, write_y=True):
        """compulsory HBM traffic of one launch: entries*8 + flush records*12 + streams*8 + X read once
        + Y written once (+ one read and one write of the fused accumulator); pads not counted"""
        b = self.nnz * 8 + self.n_flush * 12 + self.n_blocks * SWEPT_WAVES * 8 + self.n_cols * self.d * 4
        if write_y:
            b += self.n_rows * self.d * 4
        if acc:
            b += 2 * self.n_rows * self.d * 4
        return b


class CsrPlan:
    """One sparse matrix (n_rows x n_cols) as work streams: the d-independent part (row segments
    dealt to streams) lives here, `packed(d)` gives the device layout for an embedding size."""

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
        self.perm_host = perm                                           # CSR position -> original COO entry
        self._packed = {}
        self._swept = {}
        self._swept_emap = {}
        same = (share_from is not None and share_from.n_rows == self.n_rows and share_from.n_cols == self.n_cols
                and np.array_equal(share_from.rowptr_host, rowptr) and np.array_equal(share_from.csr_col_host, col)
                and np.array_equal(share_from.csr_val_host, val))
        self._alias = share_from if same else None
        if same:                                                        # symmetric matrix: reuse everything
            for k in ('r_ptr', 'r_dst', 'long_row', 'long_ptr', 'n_slots', 'n_waves', 'n_rseg', 'n_long',
                      'src_index_host', 'r_ptr_host', 'r_len_entries_host'):
                setattr(self, k, getattr(share_from, k))
            self.shared = True
        else:
            if n_waves is None:
                import os
                n_waves = min(int(os.environ.get('SSLREC_SPMM_STREAMS', MAX_WAVES)), max(1, self.nnz // MIN_STREAM))
            if seg_max is None:
                seg_max = max(64, -(-self.nnz // max(n_waves, 1)) // 2)
            (_, _, r_ptr, r_len, r_dst, src_index, long_row, long_ptr, n_slots) = \
                _streams(rowptr, n_waves, int(seg_max), row_class)
            dev = self.device
            self.src_index_host = src_index                                 # stream-sequence entry -> CSR position
            self.r_ptr_host = r_ptr.astype(np.int64)
            self.r_len_entries_host = r_len
            self.r_ptr = torch.from_numpy(r_ptr).to(dev)
            self.r_dst = torch.from_numpy(r_dst).to(dev)
            self.long_row = torch.from_numpy(long_row).to(dev)
            self.long_ptr = torch.from_numpy(long_ptr).to(dev)
            self.n_slots, self.n_waves = int(n_slots), int(r_ptr.size - 1)
            self.n_rseg, self.n_long = int(r_len.size), int(long_row.size)
            self.shared = False

    def packed(self, d):
        """device layout for embedding size d (built on first use, cached)"""
        d = int(d)
        if d not in (32, 64, 128, 256):
            raise ValueError('embedding size %d not supported by the HIP SpMM (supported: 32, 64, 128, 256)' % d)
        if d not in self._packed:
            alias = self._alias.packed(d) if self._alias is not None else None
            self._packed[d] = PackedLayout(self, d, alias_of=alias)
        return self._packed[d]

    def swept(self, d):
        """column-swept layout for embedding size d, or None when the output table does not fit the LDS
        (or SSLREC_SPMM_SWEPT=0); built on first use, cached; A^T of a symmetric matrix shares A's"""
        d = int(d)
        if d not in self._swept:
            lay = None
            if swept_enabled() and d in (32, 64, 128, 256) and self.nnz > 0 and SweptLayout.fits(self.n_rows, self.n_cols, d):
                try:
                    lay = self._alias.swept(d) if self._alias is not None else SweptLayout(self, d)
                except ValueError:          # unsuitable degree distribution: the streamed kernel takes it
                    lay = None
            self._swept[d] = lay
        return self._swept[d]

    def swept_edge_map(self, d):
        """element of the swept layout -> original COO entry (-1 for pads), on the device.  Lives on the plan:
        A^T of a symmetric matrix shares A's layout but is governed by the TRANSPOSED entries' mask bits."""
        d = int(d)
        if d not in self._swept_emap:
            lay = self.swept(d)
            em = np.full(max(lay.n_elem, 1), -1, dtype=np.int32)
            em[lay.elem_host] = self.perm_host[lay.csr_pos_host]
            self._swept_emap[d] = torch.from_numpy(em).to(self.device)
        return self._swept_emap[d]

    def algorithmic_bytes(self, d, acc=False, write_y=True):
        """compulsory HBM traffic of one streamed-kernel launch (see PackedLayout.algorithmic_bytes)"""
        b = self.nnz * 8 + self.n_rseg * 8 + self.n_waves * 16 + self.n_cols * d * 4
        if write_y:
            b += self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


class PropGraph:
    """Forward + backward plans of one adjacency, and the EdgeDrop machinery."""

    def __init__(self, rows, cols, vals, shape, device, seg_max=SEG_MAX, bipartite_split=None):
        """`bipartite_split` = number of users U when the matrix is the (U+I)^2 bipartite adjacency:
        rows < U only touch columns >= U and vice versa; the streams then walk user rows first and
        item rows second (temporal phases, see _streams)."""
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
        self.keep = keep.to(graph.device).to(torch.uint8).contiguous()      # copy first, convert on the device
        if self.keep.numel() != graph.nnz:
            raise ValueError('mask length %d != number of entries %d' % (self.keep.numel(), graph.nnz))
        self.scale = float(scale)
        self.shape = graph.shape
        self._compact = {}

    def compact(self, which, d):
        """(col, val, r_len, w_len) override arrays for plan `which` ('fwd' or 'bwd') packed for d."""
        key = (which, int(d))
        if key not in self._compact:
            lay = getattr(self.graph, which).packed(d)
            dev = lay.device
            col = torch.empty(max(lay.n_elem, 1), dtype=torch.int32, device=dev)
            val = torch.empty(max(lay.n_elem, 1), dtype=torch.float32, device=dev)
            r_len = torch.empty(max(lay.n_rseg, 1), dtype=torch.int32, device=dev)
            w_len = torch.empty(max(lay.n_waves, 1), dtype=torch.int32, device=dev)
            lib = _lib.load()
            rc = lib.sslrec_edge_drop_compact(C.byref(lay.c_struct()), lay.edge_map.data_ptr(), self.keep.data_ptr(),
                                              self.scale, col.data_ptr(), val.data_ptr(), r_len.data_ptr(),
                                              w_len.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, 'sslrec_edge_drop_compact')
            self._compact[key] = (col, val, r_len, w_len)
        return self._compact[key]

    def masked(self, which, d):
        """(pack, val, w_steps) override arrays for the column-swept layout of plan `which`: every lane group's
        stream compacted to its kept entries (sslrec_swept_compact), values rescaled"""
        key = ('swept', which, int(d))
        if key not in self._compact:
            plan = getattr(self.graph, which)
            lay = plan.swept(d)
            pack = torch.empty(max(lay.n_elem, 1), dtype=torch.int32, device=lay.device)
            val = torch.empty(max(lay.n_elem, 1), dtype=torch.float32, device=lay.device)
            steps = torch.empty(lay.n_blocks * SWEPT_WAVES, dtype=torch.int32, device=lay.device)
            rc = _lib.load().sslrec_swept_compact(C.byref(lay.c_struct()), plan.swept_edge_map(d).data_ptr(), self.keep.data_ptr(),
                                                  self.scale, pack.data_ptr(), val.data_ptr(), steps.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, 'sslrec_swept_compact')
            self._compact[key] = (pack, val, steps)
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

    def compact(self, which, d):
        key = (which, int(d))
        if key not in self._compact:
            lay = getattr(self.graph, which).packed(d)
            em = lay.edge_map.long()
            vals = torch.where(em >= 0, self.vals[em.clamp(min=0)], torch.zeros((), device=self.vals.device))
            self._compact[key] = (None, vals.contiguous(), None, None)
        return self._compact[key]

    def masked(self, which, d):
        """(None, val, None) override for the column-swept layout: the new values in element order"""
        key = ('swept', which, int(d))
        if key not in self._compact:
            em = getattr(self.graph, which).swept_edge_map(d).long()
            vals = torch.where(em >= 0, self.vals[em.clamp(min=0)], torch.zeros((), device=self.vals.device))
            self._compact[key] = (None, vals.contiguous(), None)
        return self._compact[key]

    def transposed(self):
        t = object.__new__(RevaluedView)
        t.graph, t.vals, t.shape = self.graph.transposed(), self.vals, (self.shape[1], self.shape[0])
        t._compact = {}
        return t


def graph_of(adj):
    """PropGraph of a torch sparse adjacency, built once and cached ON the tensor object -- the CSR
    conversion the reference repeats inside every spmm call.  (No cache keyed by storage address:
    a freed adjacency's address can be reused by a different one.)"""
    if isinstance(adj, (PropGraph, DroppedView, RevaluedView)):
        return adj
    g = getattr(adj, '_sslrec_graph', None)
    if g is None:
        g = PropGraph.from_torch_sparse(adj)
        try:
            adj._sslrec_graph = g
        except Exception:       # an object that cannot carry attributes: rebuilt on every call
            pass
    return g
