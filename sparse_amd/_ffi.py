"""ctypes binding of the C-ABI library ``libsparse_amd.so`` (declared in include/sparse_amd.h).

The library is the product's only compute path.  If it is missing this module raises — there
is no CPU / PyTorch fallback (a silent fallback would void every parity claim).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPAMD_LIB: load an alternative build of the same ABI (kernel-tuning experiments only)
LIB_PATH = os.environ.get("SPAMD_LIB") or os.path.join(_HERE, "_lib", "libsparse_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sparse_amd.h")

# dtype codes (include/sparse_amd.h)
F32, F64, I32, I64, BF16, U8 = 0, 1, 2, 3, 4, 5
MAX_NDIM = 16
EXACT_MULADD = 1
TILED_GROUP_ENDS = 2
TILED_INT32 = 8
SPMM_ROWGROUP = 4
SPMM_ROWVEC = 16

_lib = None


class HipBackendError(RuntimeError):
    """Raised when libsparse_amd.so is missing or a C-ABI entry point reports an error."""


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipBackendError(
                f"{LIB_PATH} not found: build it with `python -m sparse_amd.csrc.build` "
                "(there is deliberately no CPU fallback)")
        try:
            _lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise HipBackendError(f"cannot load {LIB_PATH}: {e}") from e
        _declare(_lib)
    return _lib


_C = ctypes
_i64, _int, _vp, _u32 = _C.c_int64, _C.c_int, _C.c_void_p, _C.c_uint

# name -> (restype, argtypes).  Kept in lock-step with include/sparse_amd.h; the
# `-m "not gpu"` test suite parses the header and checks both the table and the .so.
SIGNATURES = {
    "spamd_version": (_int, []),
    "spamd_target_arch": (_C.c_char_p, []),
    "spamd_coo_linearize": (_int, [_int, _int, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "spamd_coo_delinearize": (_int, [_int, _int, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "spamd_permute_keys": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_keys_check": (_int, [_i64, _vp, _vp, _vp]),
    "spamd_coo_broadcast": (_int, [_int, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "spamd_dense_nonfill_work_words": (_i64, [_i64]),
    "spamd_dense_nonfill": (_int, [_int, _i64, _vp, _C.c_uint64, _C.c_uint64, _vp, _vp, _vp, _vp]),
    "spamd_keys_lead_last_limits": (_i64, [_int]),
    "spamd_keys_lead_last": (_int, [_int, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "spamd_coords_check": (_int, [_int, _int, _i64, _vp, _i64, _vp, _vp, _vp]),
    "spamd_flag_heads": (_int, [_i64, _vp, _vp, _vp]),
    "spamd_flag_ne_bits": (_int, [_int, _i64, _vp, _C.c_uint64, _C.c_uint64, _vp, _vp]),
    "spamd_count_eq_bits": (_int, [_int, _i64, _vp, _C.c_uint64, _C.c_uint64, _vp, _vp]),
    "spamd_compact": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _vp]),
    "spamd_gather": (_int, [_int, _i64, _vp, _vp, _vp, _vp]),
    "spamd_scatter": (_int, [_int, _i64, _vp, _vp, _vp, _vp]),
    "spamd_compact_rows": (_int, [_int, _int, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "spamd_gather_rows": (_int, [_int, _int, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "spamd_keys_to_csr": (_int, [_int, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "spamd_csr_to_keys": (_int, [_int, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "spamd_rows_to_indptr": (_int, [_int, _i64, _vp, _i64, _vp, _vp]),
    "spamd_csx_swap_ws_bytes": (_i64, [_i64]),
    "spamd_csx_swap": (_int, [_int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "spamd_csx_swap8_ws_bytes": (_i64, [_i64]),
    "spamd_csx_swap8": (_int, [_int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "spamd_sort_pairs_ws_bytes": (_i64, [_i64]),
    "spamd_sort_pairs": (_int, [_i64, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp]),
    "spamd_sort_kv_ws_bytes": (_i64, [_int, _i64]),
    "spamd_sort_kv": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp]),
    "spamd_iota": (_int, [_i64, _vp, _vp]),
    "spamd_scan_ws_bytes": (_i64, [_i64]),
    "spamd_exclusive_scan": (_int, [_i64, _vp, _vp, _vp, _i64, _vp]),
    "spamd_convert": (_int, [_int, _int, _i64, _vp, _vp, _vp]),
    "spamd_lower_bound_match": (_int, [_i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "spamd_invert_flags": (_int, [_i64, _vp, _vp, _vp]),
    "spamd_union_positions": (_int, [_i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_fill": (_int, [_int, _i64, _vp, _C.c_uint64, _vp]),
    "spamd_ewise_binary": (_int, [_int, _int, _i64, _vp, _int, _vp, _int, _vp, _vp]),
    "spamd_ewise_unary": (_int, [_int, _int, _i64, _vp, _vp, _vp]),
    "spamd_ewise_select": (_int, [_int, _i64, _vp, _vp, _int, _vp, _int, _vp, _vp]),
    "spamd_segment_reduce": (_int, [_int, _int, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_count": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_expand": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "spamd_sddmm": (_int, [_int, _int, _int, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "spamd_sddmm_has_panels": (_int, [_int, _i64]),
    "spamd_sddmm_panel_keys": (_int, [_int, _i64, _vp, _i64, _i64, _vp, _vp]),
    "spamd_sddmm_panel_row_bytes": (_i64, [_int, _i64]),
    "spamd_sddmm_panels": (_int, [_int, _int, _int, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp]),
    "spamd_sddmm_tile_size": (_int, []),
    "spamd_sddmm_tile_keys": (_int, [_int, _i64, _vp, _vp, _i64, _vp, _vp]),
    "spamd_sddmm_tile_classify": (_int, [_i64, _vp, _i64, _vp, _vp, _vp]),
    "spamd_sddmm_mfma_tiles": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64,
                                      _i64, _vp, _vp]),
    "spamd_merge_num_blocks": (_i64, [_i64, _i64]),
    "spamd_merge_partition": (_int, [_i64, _vp, _i64, _vp, _vp, _vp]),
    "spamd_merge_union": (_int, [_int, _int, _int, _i64, _vp, _vp, _i64, _vp, _vp, _C.c_uint64, _C.c_uint64,
                                 _C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_merge_fused_blocks": (_i64, [_i64, _i64]),
    "spamd_merge_union_fused": (_int, [_int, _int, _i64, _vp, _vp, _i64, _vp, _vp, _C.c_uint64, _C.c_uint64, _C.c_uint64,
                                       _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_row_products": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_rows_capacity": (_i64, [_int, _i64, _i64]),
    "spamd_spgemm_rows": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_small_max_cols": (_i64, [_int]),
    "spamd_spgemm_small": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_bitmap_limits": (_i64, [_int, _int]),
    "spamd_spgemm_bitmap": (_int, [_int, _int, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_classify_rows": (_int, [_int, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "spamd_spgemm_unpack": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spgemm_pack": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_reduce_fill": (_int, [_int, _int, _i64, _vp, _vp, _i64, _C.c_double, _i64, _vp]),
    "spamd_reduce_fill_count": (_int, [_int, _int, _i64, _vp, _vp, _vp, _i64, _C.c_double, _i64, _C.c_uint64, _vp, _vp]),
    "spamd_group_reduce_ws_bytes": (_i64, [_int, _i64]),
    "spamd_group_reduce": (_int, [_int, _int, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "spamd_hot_rows_combine": (_int, [_int, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "spamd_reduce_all_ws_bytes": (_i64, []),
    "spamd_reduce_all": (_int, [_int, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "spamd_spmm_tiled_params": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_count": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_fill": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "spamd_spmm_tiled_inspect": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_inspect_csc_ws": (_i64, [_i64, _i64]),
    "spamd_spmm_tiled_inspect_csc": (_int, [_int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_map_stats": (_int, [_int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_map_groups": (_i64, [_vp]),
    "spamd_spmm_tiled_map_build": (_int, [_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_inspect_mapped": (_int, [_int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "spamd_spmm_tiled_mapped": (_int, [_int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _u32, _vp]),
    "spamd_spmm_tiled_keys": (_int, [_i64, _vp, _i64, _vp, _vp]),
    "spamd_spmm_tiled_lists": (_int, [_int, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "spamd_spmm_tiled_pack": (_int, [_int, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "spamd_spmm_tiled": (_int, [_int, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _u32, _vp]),
    "spamd_has_nan": (_int, [_int, _i64, _vp, _vp, _vp]),
    "spamd_has_nan_async": (_int, [_int, _i64, _vp, _vp, _vp]),
    "spamd_spmm_csr": (_int, [_int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _u32, _vp]),
    "spamd_spmm_csr_ldsb": (_int, [_int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _u32, _vp]),
    "spamd_deliver_words": (_int, [_vp, _int, _vp, _i64, _vp]),
    "spamd_transpose_2d": (_int, [_int, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "spamd_spmm_csr_stream": (_int, [_int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _u32, _vp]),
    "spamd_spmm_csr_stream_fits": (_int, [_int, _i64, _i64, _i64, _vp, _vp]),
    "spamd_spmm_csr_ldsb_fits": (_int, [_int, _i64, _i64, _i64, _vp, _i64, _vp, _i64]),
}


def _declare(handle):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here == symbol missing: loud by design
        fn.restype = res
        fn.argtypes = args


def header_symbols(path=HEADER_PATH):
    """Names of every function declared in include/sparse_amd.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spamd_[a-z0-9_]+)\s*\(", text)))


def check(code, what):
    if code != 0:
        kind = {-1: "invalid argument", -2: "unsupported dtype", -3: "workspace too small"}.get(
            code, f"hipError_t {code}" if code > 0 else f"error {code}")
        err = HipBackendError(f"{what} failed: {kind}")
        err.code = code      # > 0: a hipError_t (a device fault, not a declined argument); < 0: SPAMD_E*
        raise err


CALLS = 0   # C-ABI calls made so far (a cheap counter: the benches report calls per operation from its differences)


def call(name, *args):
    global CALLS
    CALLS += 1
    check(getattr(lib(), name)(*args), name)
