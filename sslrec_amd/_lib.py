"""ctypes binding of libsslrec_hip.so (C ABI declared in include/sslrec_hip.h).

There is deliberately NO fallback: if the shared object is missing or a symbol is
absent, importing the ops fails loudly.  Build with `python __graft_entry__.py` (or
`make -C sslrec_amd/csrc`).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SO_PATH = os.environ.get('SSLREC_HIP_LIBRARY') or os.path.join(CSRC, 'libsslrec_hip.so')      # override: kernel experiments

E_BADARG = 1001
EXPECTED_ABI = 7          # SSLREC_ABI_VERSION of include/sslrec_hip.h these bindings were written against


class CsrStruct(C.Structure):
    """mirror of sslrec_csr_t"""
    _fields_ = [
        ('n_rows', C.c_int32), ('n_cols', C.c_int32), ('nnz', C.c_int32), ('d', C.c_int32), ('n_elem', C.c_int32),
        ('col', C.c_void_p), ('val', C.c_void_p),
        ('n_waves', C.c_int32),
        ('w_start', C.c_void_p), ('w_len', C.c_void_p), ('r_ptr', C.c_void_p),
        ('n_rseg', C.c_int32),
        ('r_len', C.c_void_p), ('r_dst', C.c_void_p),
        ('n_long', C.c_int32),
        ('long_row', C.c_void_p), ('long_ptr', C.c_void_p),
        ('n_slots', C.c_int32),
    ]


class BundledStruct(C.Structure):
    """mirror of sslrec_bundled_t"""
    _fields_ = [
        ('n_rows', C.c_int32), ('n_cols', C.c_int32), ('nnz', C.c_int32), ('d', C.c_int32), ('n_elem', C.c_int32),
        ('col', C.c_void_p), ('val', C.c_void_p),
        ('n_waves', C.c_int32),
        ('w_start', C.c_void_p), ('w_ptr', C.c_void_p),
        ('n_bundles', C.c_int32),
        ('b_steps', C.c_void_p), ('b_dst', C.c_void_p),
        ('n_long', C.c_int32),
        ('long_row', C.c_void_p), ('long_ptr', C.c_void_p),
        ('n_slots', C.c_int32),
    ]


class SweptStruct(C.Structure):
    """mirror of sslrec_swept_t"""
    _fields_ = [
        ('n_rows', C.c_int32), ('n_cols', C.c_int32), ('nnz', C.c_int32), ('d', C.c_int32),
        ('n_elem', C.c_int32), ('n_blocks', C.c_int32), ('n_slots', C.c_int32),
        ('pack', C.c_void_p), ('val', C.c_void_p),
        ('w_start', C.c_void_p), ('w_steps', C.c_void_p),
        ('wf_ptr', C.c_void_p), ('cf_ptr', C.c_void_p), ('f_row', C.c_void_p), ('f_start', C.c_void_p), ('f_n', C.c_void_p),
    ]


class PlanInfoStruct(C.Structure):
    """mirror of sslrec_plan_info_t"""
    _fields_ = [('n_rows', C.c_int32), ('n_cols', C.c_int32), ('nnz', C.c_int64),
                ('kind', C.c_int32), ('d', C.c_int32), ('xcd_split', C.c_int32),
                ('n_elem', C.c_int32), ('n_blocks', C.c_int32), ('n_slots', C.c_int32), ('n_streams', C.c_int32),
                ('n_rseg', C.c_int32), ('n_long', C.c_int32), ('xcd_col_pairs', C.c_int64)]


class EpilogueViewsStruct(C.Structure):
    """mirror of sslrec_epilogue_views_t"""
    _fields_ = [('n_views', C.c_int32), ('eps', C.c_float), ('Y', C.c_void_p * 4), ('noise', C.c_void_p * 4),
                ('acc_in', C.c_void_p * 4), ('acc_out', C.c_void_p * 4),
                ('philox', C.c_void_p), ('philox_stream', C.c_uint32 * 4), ('philox_noise', C.c_int32 * 4),
                ('row_scale', C.c_void_p), ('scale_flags', C.c_int32)]


class EpilogueStruct(C.Structure):
    """mirror of sslrec_epilogue_t"""
    _fields_ = [('noise', C.c_void_p), ('eps', C.c_float), ('acc_in', C.c_void_p), ('acc_out', C.c_void_p),
                ('philox', C.c_void_p), ('philox_stream', C.c_uint32),
                ('noise_sumsq', C.c_void_p), ('noise_row_stride', C.c_int32), ('noise_col_off', C.c_int32),
                ('axpy_x', C.c_void_p), ('axpy_alpha', C.c_float), ('axpy_scale', C.c_void_p), ('x_row_bits', C.c_void_p),
                ('n_sum_in', C.c_int32), ('sum_in', C.c_void_p * 3),
                ('row_scale', C.c_void_p), ('scale_flags', C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float

# name -> (restype, argtypes); must list EVERY symbol include/sslrec_hip.h declares
SIGNATURES = {
    'sslrec_abi_version': (C.c_int, []),
    'sslrec_row_bits3': (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, _I, _I, _P, _P]),
    'sslrec_debug_stamp_next_launch': (C.c_int, [C.c_void_p]),
    'sslrec_debug_wall_clock_khz': (C.c_int, []),
    'sslrec_philox_row_sumsq': (C.c_int, [_P, C.c_uint32, _I, _I, _P, _P]),
    'sslrec_row_sumsq_f32': (C.c_int, [_P, _I, _I, _P, _P]),
    'sslrec_debug_gather_rows': (C.c_int, [_P, C.c_uint32, _I, _I, _I, _P, _P, _P]),
    'sslrec_debug_swept_trace': (C.c_int, [C.c_int, _P, C.c_int]),
    'sslrec_spmm_csr_f32': (C.c_int, [C.POINTER(CsrStruct), _P, _P, _P, _P, _P, _I, _P, C.POINTER(EpilogueStruct), _P, _P]),
    'sslrec_spmm_bundled_f32': (C.c_int, [C.POINTER(BundledStruct), _P, _P, _I, _P, C.POINTER(EpilogueStruct), _P, _P]),
    'sslrec_spmm_swept_f32': (C.c_int, [C.POINTER(SweptStruct), _P, _P, _P, _P, _I, _P, C.POINTER(EpilogueStruct), _P]),
    'sslrec_swept_deferred_sum_ok': (C.c_int, [C.POINTER(SweptStruct)]),
    'sslrec_spmm_swept_views_f32': (C.c_int, [C.POINTER(SweptStruct), _P, _I, C.POINTER(EpilogueViewsStruct), _P]),
    'sslrec_swept_compact': (C.c_int, [C.POINTER(SweptStruct), _P, _P, _F, _P, _P, _P, _P]),
    'sslrec_edge_drop_compact': (C.c_int, [C.POINTER(CsrStruct), _P, _P, _F, _P, _P, _P, _P, _P]),
    'sslrec_full_predict_f32': (C.c_int, [_P, _P, _I, _P, _I, _I, _P, _I, _P, _P]),
    'sslrec_eval_topk_ws_bytes': (C.c_size_t, [_I, _I, _I]),
    'sslrec_eval_topk_f32': (C.c_int, [_P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    'sslrec_sample_negs': (C.c_int, [_P, C.c_int64, _P, _P, _I, _P, C.c_uint32, _P, _P]),
    'sslrec_sample_negs_mt19937': (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _I, _I, _P, _P]),
    'sslrec_philox_advance': (C.c_int, [_P, _P]),
    'sslrec_philox_fill_f32': (C.c_int, [_P, C.c_uint32, _P, C.c_size_t, _P]),
    'sslrec_mt19937_uniform_f32': (C.c_int, [_P, _P, C.c_int64, _P]),
    'sslrec_mt19937_keep_mask': (C.c_int, [_P, C.c_float, _P, C.c_int64, _P]),
    'sslrec_mt19937_jump_poly': (C.c_int, [_P, _P, _P, _P]),
    'sslrec_mt19937_par_ws_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'sslrec_mt19937_uniform_par_f32': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, _P, _P, C.c_int64, _P]),
    'sslrec_mt19937_keep_mask_par': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, _P, C.c_float, _P, C.c_int64, _P]),
    'sslrec_swept_compact_philox': (C.c_int, [C.POINTER(SweptStruct), _P, _F, _P, C.c_uint32, _F, _P, _P, _P, _P]),
    'sslrec_edge_drop_compact_philox': (C.c_int, [C.POINTER(CsrStruct), _P, _F, _P, C.c_uint32, _F, _P, _P, _P, _P, _P]),
    'sslrec_bundled_drop_values': (C.c_int, [C.POINTER(BundledStruct), _P, _P, _F, _P, C.c_uint32, _F, _P, _P]),
    'sslrec_bundled_compact': (C.c_int, [C.POINTER(BundledStruct), _P, _P, _F, _P, _P, _P, _P, _P]),
    'sslrec_bundled_compact_philox': (C.c_int, [C.POINTER(BundledStruct), _P, _F, _P, C.c_uint32, _F, _P, _P, _P, _P, _P]),
    'sslrec_spmm_bundled_view_f32': (C.c_int, [C.POINTER(BundledStruct), _P, _P, _P, _P, _P, _I, _P, C.POINTER(EpilogueStruct), _P, _P]),
    'sslrec_plan_build_coo': (C.c_int, [_P, _P, _P, C.c_int64, _I, _I, C.POINTER(C.c_void_p)]),
    'sslrec_plan_build_csr': (C.c_int, [_P, _P, _P, _I, _I, C.POINTER(C.c_void_p)]),
    'sslrec_plan_set_option': (C.c_int, [_P, C.c_char_p, C.c_int64]),
    'sslrec_plan_layout': (C.c_int, [_P, _I, _I, _I]),
    'sslrec_plan_info': (C.c_int, [_P, _I, _I, C.POINTER(PlanInfoStruct)]),
    'sslrec_plan_host_array': (C.c_int, [_P, _I, _I, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'sslrec_plan_upload': (C.c_int, [_P, _I, _I, _P]),
    'sslrec_plan_swept': (C.c_void_p, [_P, _I]),
    'sslrec_plan_csr': (C.c_void_p, [_P, _I]),
    'sslrec_plan_bundled': (C.c_void_p, [_P, _I]),
    'sslrec_plan_edge_map': (C.c_void_p, [_P, _I, _I]),
    'sslrec_plan_spmm_f32': (C.c_int, [_P, _I, _P, _P, C.POINTER(EpilogueStruct), _P]),
    'sslrec_plan_free': (None, [_P]),
    'sslrec_bpr_ws_bytes': (C.c_size_t, [_I]),
    'sslrec_bpr_fwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    'sslrec_bpr_fwd_total_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    'sslrec_bpr_bwd_ws_bytes': (C.c_size_t, [_I, _I]),
    'sslrec_scatter_ws_bytes': (C.c_size_t, [_I]),
    'sslrec_bpr_bwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    'sslrec_bpr_bwd_table_init': (C.c_int, [_P, _I, _I, _P]),
    'sslrec_bpr_bwd_kept_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    'sslrec_infonce_ws_bytes': (C.c_size_t, [_I, _I, _I]),
    'sslrec_infonce_fwd_f32': (C.c_int, [_P, _P, _P, _P, _I, _P, _I, _I, _F, _I, _P, _P, _P]),
    'sslrec_infonce_bwd_f32': (C.c_int, [_P, _P, _P, _P, _I, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P]),
    'sslrec_infonce_bwd_scatter_f32': (C.c_int, [_P, _P, _P, _P, _I, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'sslrec_infonce_shard_rowsum_f32': (C.c_int, [_P, _P, _P, _P, _I, _P, _I, _I, _F, _I, _P, _P, _P]),
    'sslrec_infonce_shard_loss_f32': (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _P]),
    'sslrec_infonce_shard_bwd_f32': (C.c_int, [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P]),
    'sslrec_infonce_shard_finish_bwd_f32': (C.c_int, [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P]),
    'sslrec_adam_tick': (C.c_int, [_P, C.c_double, C.c_double, C.c_double, _P]),
    'sslrec_adam_apply_f32': (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    'sslrec_rankq_ws_bytes': (C.c_size_t, [_I, _I]),
    'sslrec_rankq_reduce_f32': (C.c_int, [_P, C.c_int64, C.c_int64, _P, _I, _I, _I, _P, _P, _P]),
    'sslrec_rankq_expand_f32': (C.c_int, [_P, C.c_int64, C.c_int64, _P, _I, _I, _I, _P, _P]),
    'sslrec_scatter_add_rows_f32': (C.c_int, [_P, _P, _I, _I, _P, _P, _P]),
    'sslrec_sumsq_ws_bytes': (C.c_size_t, []),
    'sslrec_sumsq_fwd_f32': (C.c_int, [_P, C.c_size_t, _F, _P, _P, _P]),
    'sslrec_sumsq_bwd_f32': (C.c_int, [_P, C.c_size_t, _F, _P, _P, _P]),
    'sslrec_weighted_sum4_f32': (C.c_int, [_P, _F, _P, _F, _P, _F, _P, _F, _P, _P]),
    'sslrec_scalar_scale2_f32': (C.c_int, [_P, _F, _F, _P, _P]),
    'sslrec_add_tables_f32': (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P]),
}

_lib = None


def build(force=False):
    """Compile the HIP sources for gfx950 into sslrec_amd/csrc/libsslrec_hip.so."""
    if force and os.path.exists(SO_PATH):
        os.remove(SO_PATH)
    subprocess.run(['make', '-C', CSRC], check=True)
    return SO_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            'libsslrec_hip.so is not built (%s). Run `python __graft_entry__.py` or `make -C sslrec_amd/csrc`. '
            'sslrec_amd has no CPU / PyTorch fallback for its kernels by design.' % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    got = lib.sslrec_abi_version()
    if got != EXPECTED_ABI:              # same symbols, different argument lists: calling on would pass misaligned arguments
        raise RuntimeError('%s implements ABI version %d, these bindings expect %d: rebuild the library '
                           '(`make -C sslrec_amd/csrc` or `python __graft_entry__.py`)' % (SO_PATH, got, EXPECTED_ABI))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = 'bad argument' if rc == E_BADARG else 'hipError_t %d' % rc
        raise RuntimeError('%s failed: %s' % (what, kind))
