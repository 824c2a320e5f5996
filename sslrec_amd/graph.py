"""CSR work-list plans for the HIP SpMM (host side, numpy + torch for device memory).

The reference hands its models a `torch.sparse` COO adjacency whose entries are in an
arbitrary (in practice column-major) order and re-coalesces it inside every `t.spmm`
call.  Here the COO is converted ONCE into

  * a CSR of A   (forward  Y = A X)         and
  * a CSR of A^T (backward dX = A^T dY),

each with its device layouts (see include/sslrec_hip.h) and an `edge_map` (layout element ->
original COO entry) so the reference's per-entry EdgeDrop mask (models/aug_utils.py:28, drawn in
COO order) can be applied without rebuilding anything.  The layouts are built by the NATIVE plan
builder inside libsslrec_hip.so (sslrec_amd/csrc/plan.cpp); this module wraps them.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

SEG_MAX = None         # chunk cap for long rows of the streamed layout; None = half the mean stream length (>= 64)
SWEPT_WAVES = 16       # 1024-thread workgroups of the column-swept kernel
KIND_AUTO, KIND_SWEPT, KIND_STREAMED, KIND_BUNDLED = 0, 1, 2, 3
FLAG_NO_XCD_SPLIT = 1
# embedding sizes of the column-swept kernel; 8 and 16 exist for feature-sliced tables (a GPU holds d / P columns of every
# row, sslrec_amd/feature_shard.py); beyond the swept layout's size limits they run on the row-bundled streamed layout
SWEPT_DIMS = (8, 16, 32, 64, 128, 256)
BUNDLED_DIMS = (8, 16)


def swept_enabled():
    import os
    return os.environ.get('SSLREC_SPMM_SWEPT', '1') != '0'


def bundled_compact_enabled():
    import os
    return os.environ.get('SSLREC_BUNDLED_COMPACT', '1') != '0'


def xcd_split_enabled():
    import os
    return os.environ.get('SSLREC_SPMM_XCD_SPLIT', '1') != '0'


class _NativePlan:
    """owner of one sslrec_plan_t (sslrec_amd/csrc/plan.cpp): the CSR of a matrix and its layouts, built in C++"""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def from_coo(cls, rows, cols, vals, n_rows, n_cols):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.float32)
        h = C.c_void_p()
        rc = _lib.load().sslrec_plan_build_coo(rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, rows.size, int(n_rows),
                                               int(n_cols), C.byref(h))
        _lib.check(rc, 'sslrec_plan_build_coo')
        return cls(h)

    @classmethod
    def from_csr(cls, rowptr, col, val, n_rows, n_cols):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        h = C.c_void_p()
        rc = _lib.load().sslrec_plan_build_csr(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, int(n_rows), int(n_cols),
                                               C.byref(h))
        _lib.check(rc, 'sslrec_plan_build_csr')
        return cls(h)

    def set_option(self, name, value):
        _lib.check(_lib.load().sslrec_plan_set_option(self.handle, name.encode(), int(value)), 'sslrec_plan_set_option')

    def layout(self, d, kind, flags=0):
        """kind built (> 0) or a negative error (e.g. the swept layout does not fit)"""
        return _lib.load().sslrec_plan_layout(self.handle, int(d), int(kind), int(flags))

    def array(self, d, kind, name):
        """copy of a named host array (d = 0: the CSR itself)"""
        ptr, cnt, eb = C.c_void_p(), C.c_int64(), C.c_int32()
        rc = _lib.load().sslrec_plan_host_array(self.handle, int(d), int(kind), name.encode(), C.byref(ptr), C.byref(cnt), C.byref(eb))
        _lib.check(rc, 'sslrec_plan_host_array(%s)' % name)
        dt = {('val',): np.float32}.get((name,), np.int64 if eb.value == 8 else np.int32)
        if cnt.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (cnt.value * eb.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def info(self, d, kind):
        inf = _lib.PlanInfoStruct()
        _lib.check(_lib.load().sslrec_plan_info(self.handle, int(d), int(kind), C.byref(inf)), 'sslrec_plan_info')
        return inf

    def __del__(self):
        try:
            if self.handle:
                _lib.load().sslrec_plan_free(self.handle)
                self.handle = None
        except Exception:
            pass


def _edge_map(native, d, kind, perm_outer):
    """element -> ORIGINAL entry: the native map points at the entries the plan was built from; when those were a
    re-labelled CSR (shards), `perm_outer` takes them back to the caller's COO order"""
    em = native.array(d, kind, 'edge_map')
    if perm_outer is not None:
        em = np.where(em >= 0, perm_outer[np.maximum(em, 0)], -1).astype(np.int32)
    return em


class PackedLayout:
    """Device arrays of one streamed CSR packed for ONE embedding size d (`sslrec_csr_t`), built natively."""

    def __init__(self, plan, d):
        nat = plan.native
        inf = nat.info(d, KIND_STREAMED)
        self.d, self.G = int(d), 256 // int(d)
        self.n_rows, self.n_cols, self.nnz, self.device = plan.n_rows, plan.n_cols, plan.nnz, plan.device
        self.n_waves, self.n_rseg, self.n_long, self.n_slots, self.n_elem = inf.n_streams, inf.n_rseg, inf.n_long, inf.n_slots, inf.n_elem
        dev = self.device
        t = lambda name: torch.from_numpy(nat.array(d, KIND_STREAMED, name)).to(dev)
        self.col, self.val = t('col'), t('val')
        self.w_start, self.w_len, self.r_ptr = t('w_start'), t('w_len'), t('r_ptr')
        self.r_len, self.r_dst = t('r_len'), t('r_dst')
        self.long_row, self.long_ptr = t('long_row'), t('long_ptr')
        self.edge_map = torch.from_numpy(_edge_map(nat, d, KIND_STREAMED, plan.perm_outer)).to(dev)   # element -> original COO entry
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
        """Compulsory HBM traffic of one SpMM, SURVEY.md §8d's formula as written: entries * (4 + 4) + (n_rows + 1) * 4 + X read
        once + Y written once (+ one read and one write of the fused accumulator).  The layout's own metadata (row segments,
        streams) and its pads are NOT counted: these are the algorithm's bytes, not the layout's."""
        d = self.d
        b = self.nnz * 8 + (self.n_rows + 1) * 4 + self.n_cols * d * 4
        if write_y:
            b += self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


class BundledLayout:
    """Device arrays of the ROW-BUNDLED streamed layout (`sslrec_bundled_t`, spmm_bundle_kernel in csrc/spmm.hip) for a
    narrow embedding size (8 / 16; 32 with SSLREC_SPMM_BUNDLED32=1): every lane group owns a whole output row for the
    length of a bundle of G = 256/d rows of similar length.  The edge map (4 bytes per element: 1.4 GB for config 5's
    adjacency) is only uploaded when a view asks for it."""

    def __init__(self, plan, d):
        nat = plan.native
        inf = nat.info(d, KIND_STREAMED)
        assert inf.kind == KIND_BUNDLED
        self.d, self.G = int(d), 256 // int(d)
        self.n_rows, self.n_cols, self.nnz, self.device = plan.n_rows, plan.n_cols, plan.nnz, plan.device
        self.n_waves, self.n_bundles, self.n_long, self.n_slots, self.n_elem = inf.n_streams, inf.n_rseg, inf.n_long, inf.n_slots, inf.n_elem
        dev = self.device
        t = lambda name: torch.from_numpy(nat.array(d, KIND_STREAMED, name)).to(dev)
        self.col, self.val = t('col'), t('val')
        self.w_start, self.w_ptr = t('w_start'), t('w_ptr')
        self.b_steps, self.b_dst = t('b_steps'), t('b_dst')
        self.long_row, self.long_ptr = t('long_row'), t('long_ptr')
        self._plan = plan
        self._edge_map = None
        self._struct = None
        self._partial = None

    @property
    def edge_map(self):
        if self._edge_map is None:
            self._edge_map = torch.from_numpy(_edge_map(self._plan.native, self.d, KIND_STREAMED, self._plan.perm_outer)).to(self.device)
        return self._edge_map

    def c_struct(self):
        if self._struct is None:
            s = _lib.BundledStruct()
            s.n_rows, s.n_cols, s.nnz, s.d, s.n_elem = self.n_rows, self.n_cols, self.nnz, self.d, self.n_elem
            s.col, s.val = self.col.data_ptr(), self.val.data_ptr()
            s.n_waves, s.n_bundles = self.n_waves, self.n_bundles
            s.w_start, s.w_ptr = self.w_start.data_ptr(), self.w_ptr.data_ptr()
            s.b_steps, s.b_dst = self.b_steps.data_ptr(), self.b_dst.data_ptr()
            s.n_long = self.n_long
            s.long_row, s.long_ptr = self.long_row.data_ptr(), self.long_ptr.data_ptr()
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
        """compulsory HBM traffic of one launch, SURVEY.md §8d's formula: entries*8 + (n_rows + 1)*4 + X read once + Y written once
        (+ one read and one write of the fused accumulator); the layout's bundle records and pads are not counted"""
        d = self.d
        b = self.nnz * 8 + (self.n_rows + 1) * 4 + self.n_cols * d * 4
        if write_y:
            b += self.n_rows * d * 4
        if acc:
            b += 2 * self.n_rows * d * 4
        return b


class SweptLayout:
    """Column-swept layout (`sslrec_swept_t`, kernel sslrec_amd/csrc/spmm_swept.hip) of one plan for one embedding
    size, built natively: output rows live in LDS, every lane group owns a disjoint set of them and walks its edges
    sorted by column.  Only for matrices whose OUTPUT table fits the chip's LDS (CsrPlan.swept returns None otherwise)."""

    @staticmethod
    def steps_per_block(d):
        return min(16, int(d) // 4)      # = min(16, lanes per lane group): 2 / 4 / 8 / 16 at d = 8 / 16 / 32 / >= 64

    def __init__(self, plan, d):
        nat = plan.native
        inf = nat.info(d, KIND_SWEPT)
        d = int(d)
        # `d` = the tables' embedding size; the layout itself may be built for d/2, d/4 ... columns (`width`) and run in
        # d / width embedding-column passes when the [n_rows, d] output does not fit the chip's LDS
        self.d, self.width, self.n_pass = d, int(inf.d), d // int(inf.d)
        self.G = 256 // self.width
        self.n_rows, self.n_cols, self.nnz, self.device = plan.n_rows, plan.n_cols, plan.nnz, plan.device
        self.n_elem, self.n_blocks, self.n_slots, self.xcd_split = inf.n_elem, inf.n_blocks, inf.n_slots, bool(inf.xcd_split)
        self.xcd_col_pairs = int(inf.xcd_col_pairs)      # distinct (XCD, column) pairs: x 4 width bytes = the fabric-read floor of a launch
        dev = self.device
        t = lambda name: torch.from_numpy(nat.array(d, KIND_SWEPT, name)).to(dev)
        self.pack, self.val = t('pack'), t('val')
        self.w_start, self.w_steps = t('w_start'), t('w_steps')
        self.wf_ptr, self.cf_ptr = t('wf_ptr'), t('cf_ptr')      # flush records of every wave (one-slot rows) / workgroup (chunked rows)
        self.f_row, self.f_start, self.f_n = t('f_row'), t('f_start'), t('f_n')
        self.n_flush = int(self.n_rows)
        self.edge_map = torch.from_numpy(_edge_map(nat, d, KIND_SWEPT, plan.perm_outer)).to(dev)      # element -> original COO entry
        self._struct = None

    def c_struct(self):
        if self._struct is None:
            s = _lib.SweptStruct()
            s.n_rows, s.n_cols, s.nnz, s.d = self.n_rows, self.n_cols, self.nnz, self.width
            s.n_elem, s.n_blocks, s.n_slots = self.n_elem, self.n_blocks, self.n_slots
            s.pack, s.val = self.pack.data_ptr(), self.val.data_ptr()
            s.w_start, s.w_steps = self.w_start.data_ptr(), self.w_steps.data_ptr()
            s.wf_ptr, s.cf_ptr, s.f_row = self.wf_ptr.data_ptr(), self.cf_ptr.data_ptr(), self.f_row.data_ptr()
            s.f_start, s.f_n = self.f_start.data_ptr(), self.f_n.data_ptr()
            self._struct = s
        return self._struct

    def algorithmic_bytes(self, d=None, acc=False, write_y=True, x_rows=None, sum_in=0, pattern=False):
        """compulsory HBM traffic of one launch, SURVEY.md §8d's formula as written: entries*8 + (n_rows + 1)*4 + X read once
        + Y written once (+ one read and one write of the fused accumulator, + one read per deferred layer table `sum_in`); the
        layout's own flush records (12 bytes per row where a row pointer has 4) and its pads are NOT counted.  x_rows: a launch told
        that only so many rows of X are not zero (sslrec_epilogue_t.x_row_bits) reads those rows and a bitmap of n_cols bits.
        pattern=True: what a PATTERN launch of the factorized chain itself has to move -- no value array (entries*4), one row factor
        per output row instead; the SURVEY figure of the OPERATION (values counted) is pattern=False whatever the launch reads"""
        x_read = self.n_cols if x_rows is None else min(self.n_cols, int(x_rows))
        b = self.nnz * (4 if pattern else 8) + (self.n_rows + 1) * 4 + (self.n_rows * 4 if pattern else 0) \
            + x_read * self.d * 4 + (0 if x_rows is None else self.n_cols // 8)
        if write_y:
            b += self.n_rows * self.d * 4
        if acc:
            b += 2 * self.n_rows * self.d * 4
        b += int(sum_in) * self.n_rows * self.d * 4
        return b


class CsrPlan:
    """One sparse matrix (n_rows x n_cols): its CSR (entries sorted by row, then column; duplicates kept) and, per
    embedding size, the streamed (`packed(d)`) and column-swept (`swept(d)`) device layouts.  All of it is built by the
    native builder of libsslrec_hip.so; this class only moves the arrays into PyTorch's allocator."""

    def __init__(self, rows, cols, vals, n_rows, n_cols, device, seg_max=None, col_relabel=None):
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size >= 2 ** 31 - 1:
            raise ValueError('a single shard is limited to 2^31-1 entries (int32 CSR)')
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(rows.size)
        self.device = torch.device(device)
        nat = _NativePlan.from_coo(rows, cols, vals, n_rows, n_cols)
        self.rowptr_host = nat.array(0, 0, 'rowptr')
        self.csr_col_host, self.csr_val_host = nat.array(0, 0, 'col'), nat.array(0, 0, 'val')
        self.perm_host = nat.array(0, 0, 'perm')                           # CSR position -> original COO entry
        self.perm_outer = None
        if col_relabel is not None:      # new column ids, entry order (= summation order) unchanged
            self.csr_col_host = np.asarray(col_relabel(self.csr_col_host.astype(np.int64))).astype(np.int32)
            nat = _NativePlan.from_csr(self.rowptr_host, self.csr_col_host, self.csr_val_host, n_rows, n_cols)
            self.perm_outer = self.perm_host
        if seg_max is not None:
            nat.set_option('seg_max', int(seg_max))
        import os
        if os.environ.get('SSLREC_SWEPT_BLOCKS'):
            nat.set_option('swept_blocks', int(os.environ['SSLREC_SWEPT_BLOCKS']))
        if os.environ.get('SSLREC_XCD_BALANCE'):
            nat.set_option('xcd_balance', int(os.environ['SSLREC_XCD_BALANCE']))
        if os.environ.get('SSLREC_XCD_STAGGER'):           # experiment: the XCDs of a row class end their sweeps one after the other (plan.cpp)
            nat.set_option('xcd_stagger', int(os.environ['SSLREC_XCD_STAGGER']))
        if os.environ.get('SSLREC_XCD_CLUSTER'):           # row -> XCD co-clustering of the swept layout (plan.cpp: cocluster_rows): 0 = never,
            v = os.environ['SSLREC_XCD_CLUSTER']           # n = always, n refinement passes; auto (the builder's default) = kept when it pays
            nat.set_option('xcd_cluster', 17 if v == 'auto' else int(v))
        if os.environ.get('SSLREC_SWEPT_WIDTH'):           # widest swept layout (tests: forces embedding-column passes)
            nat.set_option('swept_width', int(os.environ['SSLREC_SWEPT_WIDTH']))
        if os.environ.get('SSLREC_SPMM_BUNDLED32'):        # 1: the streamed kind at d = 32 is the row-bundled layout
            nat.set_option('bundled32', int(os.environ['SSLREC_SPMM_BUNDLED32']))
        if os.environ.get('SSLREC_SWEPT_PASSES'):          # 0: never run the swept kernel in embedding-column passes
            nat.set_option('swept_passes', int(os.environ['SSLREC_SWEPT_PASSES']))
        self.native = nat
        self._packed = {}
        self._swept = {}

    def packed(self, d):
        """streamed device layout for embedding size d (built on first use, cached): the packed layout of
        spmm_stream_kernel at d >= 32, the row-bundled one (BundledLayout) at the narrow widths 8 / 16"""
        d = int(d)
        if d not in (8, 16, 32, 64, 128, 256):
            raise ValueError('embedding size %d not supported by the HIP SpMM (supported: 8, 16, 32, 64, 128, 256)' % d)
        if d not in self._packed:
            kind = self.native.layout(d, KIND_STREAMED)
            if kind == KIND_BUNDLED:
                self._packed[d] = BundledLayout(self, d)
            elif kind == KIND_STREAMED:
                self._packed[d] = PackedLayout(self, d)
            else:
                raise ValueError('streamed layout could not be built (int32 indexing exceeded)')
        return self._packed[d]

    def swept(self, d):
        """column-swept layout for embedding size d, or None when the output table does not fit the chip's LDS, the
        degree distribution is unsuitable (one giant row) or SSLREC_SPMM_SWEPT=0; built on first use, cached"""
        d = int(d)
        if d not in self._swept:
            lay = None
            if swept_enabled() and d in SWEPT_DIMS and self.nnz > 0:
                flags = 0 if xcd_split_enabled() else FLAG_NO_XCD_SPLIT
                if self.native.layout(d, KIND_SWEPT, flags) == KIND_SWEPT:
                    lay = SweptLayout(self, d)
            self._swept[d] = lay
        return self._swept[d]

    def swept_edge_map(self, d):
        """element of the swept layout -> original COO entry (-1 for pads), on the device"""
        return self.swept(d).edge_map

    def algorithmic_bytes(self, d, acc=False, write_y=True):
        """compulsory HBM traffic of one streamed-kernel launch (see PackedLayout.algorithmic_bytes)"""
        return self.packed(d).algorithmic_bytes(acc=acc, write_y=write_y)


class PropGraph:
    """Forward + backward plans of one adjacency, and the EdgeDrop machinery."""

    def __init__(self, rows, cols, vals, shape, device, seg_max=SEG_MAX):
        n_rows, n_cols = int(shape[0]), int(shape[1])
        self.shape = (n_rows, n_cols)
        self.device = torch.device(device)
        self.fwd = CsrPlan(rows, cols, vals, n_rows, n_cols, device, seg_max)
        self.bwd = CsrPlan(cols, rows, vals, n_cols, n_rows, device, seg_max)      # A^T: same entries, transposed
        self.nnz = self.fwd.nnz

    def factorization(self):
        """(r, c, symmetric) with value(i, j) == r[i] * c[j] for every entry, as fp32 device vectors -- or None.  The reference's
        adjacency is D^-1/2 A D^-1/2 with a binarized A (data_handler_general_cf.py:37-51, :65; D = row sums + 1e-10) and LightGCL's
        is 1 / sqrt(d_u d_i) (lightgcl.py:17-20): both factorize, and the column-swept kernel can then run a layer chain on the scaled
        table c (.) E without reading the value stream (sslrec_epilogue_t.scale_flags).  Detected, not assumed: r and c are recomputed
        from the entry counts and every stored value is checked against r[i] * c[j] (4e-7 relative: the value array holds the fp64
        product rounded once, r and c are rounded separately).  `symmetric`: square and r == c bit for bit (the chain's SCALE_Y /
        SCALE_ACC hand the next launch row_scale (.) y, which must be its column factor)."""
        if not hasattr(self, '_fact'):
            self._fact = None
            fwd = self.fwd
            if getattr(self, 'bwd', None) is not None and fwd.perm_outer is None and fwd.nnz > 0:
                deg_r = np.diff(fwd.rowptr_host).astype(np.int64)
                cols = fwd.csr_col_host.astype(np.int64)
                deg_c = np.bincount(cols, minlength=fwd.n_cols).astype(np.int64)
                r = np.where(deg_r > 0, (deg_r + 1e-10) ** -0.5, 1.0).astype(np.float32)
                c = np.where(deg_c > 0, (deg_c + 1e-10) ** -0.5, 1.0).astype(np.float32)
                prod = np.repeat(r.astype(np.float64), deg_r) * c.astype(np.float64)[cols]
                if np.all(np.abs(fwd.csr_val_host.astype(np.float64) - prod) <= 4e-7 * prod):
                    sym = fwd.n_rows == fwd.n_cols and bool(np.array_equal(r, c))
                    self._fact = (torch.from_numpy(r).to(self.device), torch.from_numpy(c).to(self.device), sym)
        return self._fact

    @classmethod
    def _single(cls, rows, cols, vals, shape, device, seg_max=SEG_MAX, col_relabel=None):
        """forward-only graph (one plan), used for the row shards of sslrec_amd.shard"""
        g = object.__new__(cls)
        g.shape = (int(shape[0]), int(shape[1]))
        g.device = torch.device(device)
        g.fwd = CsrPlan(rows, cols, vals, g.shape[0], g.shape[1], device, seg_max, col_relabel=col_relabel)
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
        f = self.factorization()
        t._fact = None if f is None else (f[1], f[0], f[2])
        return t


class DroppedView:
    """An edge-dropped view of a PropGraph: the result of EdgeDrop (aug_utils.py:18-31) without
    rebuilding the sparse tensor.  `keep` is the reference's per-COO-entry boolean mask -- or None in perf mode,
    where `philox = (PhiloxState, stream id, keep_rate)` lets the compaction kernels COMPUTE the mask bit of every
    entry (sslrec_amd/rng.py): no mask is drawn, stored or copied."""

    def __init__(self, graph, keep, scale=1.0, philox=None, entry_ids=None):
        """entry_ids (optional, int64 on the device): id of every entry of `graph` in a LARGER entry list that the mask
        / the Philox stream is defined over -- a row shard of a sharded adjacency passes the global COO ids of its entries,
        so that all ranks (and the A / A^T shards) drop the same edges (sslrec_amd/shard.py)"""
        self.graph = graph
        self.entry_ids = entry_ids
        self._emaps = {}
        if keep is None:
            if philox is None:
                raise ValueError('a DroppedView needs a keep mask or a Philox stream')
            self.keep = None
        else:
            self.keep = keep.to(graph.device).to(torch.uint8).contiguous()      # copy first, convert on the device
            if entry_ids is None and self.keep.numel() != graph.nnz:
                raise ValueError('mask length %d != number of entries %d' % (self.keep.numel(), graph.nnz))
        self.philox = philox
        self.scale = float(scale)
        self.shape = graph.shape
        self._compact = {}

    def compact(self, which, d):
        """(col, val, r_len, w_len) override arrays for plan `which` ('fwd' or 'bwd') packed for d."""
        key = (which, int(d))
        if key not in self._compact:
            lay = getattr(self.graph, which).packed(d)
            if isinstance(lay, BundledLayout) and bundled_compact_enabled():
                # narrow table beyond the swept layout: every bundle's rows compacted, the bundles moved up (sslrec_bundled_compact)
                dev = lay.device
                col = torch.empty(max(lay.n_elem, 1), dtype=torch.int32, device=dev)
                val = torch.empty(max(lay.n_elem, 1), dtype=torch.float32, device=dev)
                b_steps = torch.empty(max(lay.n_bundles, 1), dtype=torch.int32, device=dev)
                w_blocks = torch.empty(max(lay.n_waves, 1), dtype=torch.int32, device=dev)
                emap = self._edge_map(lay)
                st = torch.cuda.current_stream().cuda_stream
                if self.keep is not None:
                    rc = _lib.load().sslrec_bundled_compact(C.byref(lay.c_struct()), emap.data_ptr(), self.keep.data_ptr(), self.scale,
                                                            col.data_ptr(), val.data_ptr(), b_steps.data_ptr(), w_blocks.data_ptr(), st)
                else:
                    state, stream, keep_rate = self.philox
                    rc = _lib.load().sslrec_bundled_compact_philox(C.byref(lay.c_struct()), emap.data_ptr(), float(keep_rate), state.state.data_ptr(),
                                                                   int(stream), self.scale, col.data_ptr(), val.data_ptr(), b_steps.data_ptr(),
                                                                   w_blocks.data_ptr(), st)
                _lib.check(rc, 'sslrec_bundled_compact')
                self._compact[key] = (col, val, b_steps, w_blocks)
                return self._compact[key]
            if isinstance(lay, BundledLayout):      # SSLREC_BUNDLED_COMPACT=0 (rounds 3-4): the dropped entries get the value zero, full stream length
                val = torch.empty(max(lay.n_elem, 1), dtype=torch.float32, device=lay.device)
                emap = self._edge_map(lay)
                state, stream, keep_rate = self.philox if self.keep is None else (None, 0, 0.0)
                rc = _lib.load().sslrec_bundled_drop_values(C.byref(lay.c_struct()), emap.data_ptr(),
                                                            None if self.keep is None else self.keep.data_ptr(), float(keep_rate),
                                                            None if state is None else state.state.data_ptr(), int(stream), self.scale,
                                                            val.data_ptr(), torch.cuda.current_stream().cuda_stream)
                _lib.check(rc, 'sslrec_bundled_drop_values')
                self._compact[key] = (None, val, None, None)
                return self._compact[key]
            dev = lay.device
            col = torch.empty(max(lay.n_elem, 1), dtype=torch.int32, device=dev)
            val = torch.empty(max(lay.n_elem, 1), dtype=torch.float32, device=dev)
            r_len = torch.empty(max(lay.n_rseg, 1), dtype=torch.int32, device=dev)
            w_len = torch.empty(max(lay.n_waves, 1), dtype=torch.int32, device=dev)
            lib = _lib.load()
            st = torch.cuda.current_stream().cuda_stream
            emap = self._edge_map(lay)
            if self.keep is not None:
                rc = lib.sslrec_edge_drop_compact(C.byref(lay.c_struct()), emap.data_ptr(), self.keep.data_ptr(),
                                                  self.scale, col.data_ptr(), val.data_ptr(), r_len.data_ptr(),
                                                  w_len.data_ptr(), st)
            else:
                state, stream, keep_rate = self.philox
                rc = lib.sslrec_edge_drop_compact_philox(C.byref(lay.c_struct()), emap.data_ptr(), float(keep_rate),
                                                         state.state.data_ptr(), int(stream), self.scale, col.data_ptr(),
                                                         val.data_ptr(), r_len.data_ptr(), w_len.data_ptr(), st)
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
            lib = _lib.load()
            st = torch.cuda.current_stream().cuda_stream
            emap = self._edge_map(lay)
            if self.keep is not None:
                rc = lib.sslrec_swept_compact(C.byref(lay.c_struct()), emap.data_ptr(), self.keep.data_ptr(),
                                              self.scale, pack.data_ptr(), val.data_ptr(), steps.data_ptr(), st)
            else:
                state, stream, keep_rate = self.philox
                rc = lib.sslrec_swept_compact_philox(C.byref(lay.c_struct()), emap.data_ptr(), float(keep_rate),
                                                     state.state.data_ptr(), int(stream), self.scale, pack.data_ptr(),
                                                     val.data_ptr(), steps.data_ptr(), st)
            _lib.check(rc, 'sslrec_swept_compact')
            self._compact[key] = (pack, val, steps)
        return self._compact[key]

    def _edge_map(self, lay):
        """element -> id of the entry whose mask bit governs it"""
        if self.entry_ids is None:
            return lay.edge_map
        key = id(lay)
        if key not in self._emaps:
            em = lay.edge_map.long()
            self._emaps[key] = torch.where(em >= 0, self.entry_ids[em.clamp(min=0)], em).to(torch.int32).contiguous()
        return self._emaps[key]

    def n_kept(self):
        if self.keep is None:
            raise ValueError('the mask of a Philox view is never materialized')
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
