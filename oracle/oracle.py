"""CPU oracle for the sparse_amd hot path — TEST INFRASTRUCTURE ONLY.

Python face of ``oracle/oracle.c`` (C restatements of the reference's jitted dot kernels,
called through ctypes) plus NumPy restatements of the reference's L1 drivers
(canonicalisation, COO<->GCXS conversion, elementwise merge, grouped reduce).  Every function
cites the reference file:line it follows.  Parity status: PINNED — checked bit-for-bit
against fixtures under ``tests/golden/`` that ``oracle/gen_golden.py`` produced by running
the real reference source (``/root/reference`` under the no-op numba stub).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  ``sparse_amd`` never does.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_SUFFIX = {np.dtype("float32"): "f32", np.dtype("float64"): "f64",
           np.dtype("int32"): "i32", np.dtype("int64"): "i64"}


def use_library(path):
    """Swap the loaded shared library (bench.py's cpu_baseline leg: the -march=native build of the same source,
    oracle/build.py `build(native=True)`); returns the previous path-or-None.  `None` restores the default."""
    global _LIB
    _LIB = None if path is None else ctypes.CDLL(path)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            from . import build as _b  # builds with gcc (seconds)

            _b.build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(x):
    return ctypes.c_int64(int(x))


def dot_dtype(dt1, dt2):
    """reference _common.py:635-636"""
    return (np.zeros((), dtype=dt1) * np.zeros((), dtype=dt2)).dtype


def _prep(a_data, b, *idx):
    dtr = dot_dtype(a_data.dtype, b.dtype)
    if dtr not in _SUFFIX:
        raise TypeError(f"oracle: unsupported dtype {dtr}")
    idt = np.result_type(*[i.dtype for i in idx])
    idt = np.dtype("int32") if idt == np.dtype("int32") else np.dtype("int64")
    a_data = np.ascontiguousarray(a_data, dtype=dtr)
    b = np.ascontiguousarray(b, dtype=dtr)
    idx = [np.ascontiguousarray(i, dtype=idt) for i in idx]
    return dtr, idt, a_data, b, idx


def dot_csr_ndarray(out_shape, a_data, a_indices, a_indptr, b):
    """C restatement of `_dot_csr_ndarray` (reference _common.py:744-753)."""
    M, N = (int(s) for s in out_shape)
    dtr, idt, a_data, b, (a_indices, a_indptr) = _prep(a_data, b, a_indices, a_indptr)
    out = np.empty((M, N), dtype=dtr)
    fn = getattr(lib(), f"oracle_dot_csr_ndarray_{_SUFFIX[dtr]}_{_SUFFIX[idt]}")
    fn.restype = None
    fn(_i64(M), _i64(N), _p(a_data), _p(a_indices), _p(a_indptr), _p(b),
       _i64(b.shape[1] if b.ndim == 2 else 1), _p(out))
    return out


def dot_csc_ndarray(a_shape, b_shape, a_data, a_indices, a_indptr, b):
    """C restatement of `_dot_csc_ndarray` (reference _common.py:893-902)."""
    M, K = (int(s) for s in a_shape)
    N = int(b_shape[1])
    dtr, idt, a_data, b, (a_indices, a_indptr) = _prep(a_data, b, a_indices, a_indptr)
    out = np.empty((M, N), dtype=dtr)
    fn = getattr(lib(), f"oracle_dot_csc_ndarray_{_SUFFIX[dtr]}_{_SUFFIX[idt]}")
    fn.restype = None
    fn(_i64(M), _i64(K), _i64(N), _p(a_data), _p(a_indices), _p(a_indptr), _p(b), _i64(N), _p(out))
    return out


def dot_coo_ndarray(coords, data, b, out_shape):
    """C restatement of `_dot_coo_ndarray` (reference _common.py:999-1012).

    NOTE: takes B itself (K x N), not the transposed view the reference kernel receives.
    """
    M, N = (int(s) for s in out_shape)
    dtr, idt, data, b, (rows, cols) = _prep(data, b, coords[0], coords[1])
    out = np.empty((M, N), dtype=dtr)
    fn = getattr(lib(), f"oracle_dot_coo_ndarray_{_SUFFIX[dtr]}_{_SUFFIX[idt]}")
    fn.restype = None
    fn(_i64(len(data)), _p(rows), _p(cols), _p(data), _p(b), _i64(N), _i64(M), _i64(N), _p(out))
    return out


def dot_csr_csr(out_shape, a_data, b_data, a_indices, b_indices, a_indptr, b_indptr):
    """C restatement of `_dot_csr_csr` + `_csr_csr_count_nnz` (reference _common.py:559-570,
    666-715).  Returns (data, indices[intp], indptr[intp]) in the reference's own (unsorted,
    reverse-discovery) per-row order, explicit zeros included."""
    n_row, n_col = (int(s) for s in out_shape)
    dtr = dot_dtype(a_data.dtype, b_data.dtype)
    idt = np.result_type(a_indices.dtype, b_indices.dtype, a_indptr.dtype, b_indptr.dtype)
    idt = np.dtype("int32") if idt == np.dtype("int32") else np.dtype("int64")
    a_data = np.ascontiguousarray(a_data, dtype=dtr)
    b_data = np.ascontiguousarray(b_data, dtype=dtr)
    ai, bi, ap, bp = (np.ascontiguousarray(x, dtype=idt) for x in (a_indices, b_indices, a_indptr, b_indptr))
    cnt = getattr(lib(), f"oracle_csr_csr_count_nnz_{_SUFFIX[idt]}")
    cnt.restype = ctypes.c_int64
    nnz = cnt(_i64(n_row), _i64(n_col), _p(ai), _p(bi), _p(ap), _p(bp), None)
    data = np.empty(nnz, dtype=dtr)
    indices = np.empty(nnz, dtype=np.int64)
    indptr = np.empty(n_row + 1, dtype=np.int64)
    fn = getattr(lib(), f"oracle_dot_csr_csr_{_SUFFIX[dtr]}_{_SUFFIX[idt]}")
    fn.restype = ctypes.c_int64
    got = fn(_i64(n_row), _i64(n_col), _p(a_data), _p(b_data), _p(ai), _p(bi), _p(ap), _p(bp),
             _p(data), _p(indices), _p(indptr), _i64(nnz))
    assert got == nnz
    return data, indices, indptr


def match_arrays(a, b):
    """C restatement of `_match_arrays` (reference _umath.py:70-92)."""
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    fn = lib().oracle_match_arrays
    fn.restype = ctypes.c_int64
    n = fn(_p(a), _i64(len(a)), _p(b), _i64(len(b)), None, None)
    ai = np.empty(n, dtype=np.int64)
    bi = np.empty(n, dtype=np.int64)
    fn(_p(a), _i64(len(a)), _p(b), _i64(len(b)), _p(ai), _p(bi))
    return ai.astype(np.uintp), bi.astype(np.uintp)


# ----------------------------------------------------------------------------------------
# NumPy restatements of the L1 drivers (integer side; must match bit-exactly)
# ----------------------------------------------------------------------------------------

def linear_loc(coords, shape):
    """reference _coo/common.py:56-64 — C-order ravel of coords."""
    if len(shape) == 0 or coords.shape[0] == 0:
        return np.zeros(coords.shape[1], dtype=np.intp)
    return np.ravel_multi_index(tuple(np.asarray(c, dtype=np.int64) for c in coords), shape)


def equivalent_bits(x, fill):
    """reference _utils.py:406-452 without `loose`: bit-wise equality (-0.0 != 0.0)."""
    return _bits_equal(np.asarray(x), fill)


def coo_canonicalize(coords, data, shape, sum_duplicates=True, prune=False, fill_value=0):
    """COO canonical form, Appendix D1 (reference _coo/core.py:1294-1371):
    stable sort by C-order key, sum duplicates with reduceat in the data dtype, optional
    bit-wise prune of fill values."""
    coords = np.asarray(coords)
    data = np.asarray(data)
    lin = linear_loc(coords, shape)
    if (np.diff(lin) < 0).any():
        order = np.argsort(lin, kind="mergesort")
        coords, data, lin = coords[:, order], data[order], lin[order]
    if sum_duplicates and len(lin) > 0 and (np.diff(lin) == 0).any():
        heads = np.concatenate(([True], lin[1:] != lin[:-1]))
        idx = np.flatnonzero(heads)
        data = np.add.reduceat(data, idx).astype(data.dtype)
        coords = coords[:, idx]
    if prune and len(data):
        keep = ~np.asarray(equivalent_bits(data, fill_value))
        coords, data = coords[:, keep], data[keep]
    return coords, data


def coo_to_gcxs(coords, data, shape, compressed_axes=None):
    """Appendix D2 (reference _compressed/compressed.py:25-77)."""
    ndim = len(shape)
    if ndim == 1:
        return data.copy(), np.asarray(coords[0]).copy(), np.empty(0, dtype=np.int64), None
    if compressed_axes is None:
        compressed_axes = (int(np.argmin(shape)),)
    compressed_axes = tuple(int(c) for c in compressed_axes)
    axis_order = list(compressed_axes) + [a for a in range(ndim) if a not in compressed_axes]
    rshape = tuple(shape[a] for a in axis_order)
    R = int(np.prod(rshape[: len(compressed_axes)], dtype=np.int64))
    C = int(np.prod(rshape[len(compressed_axes):], dtype=np.int64))
    lin = linear_loc(np.asarray(coords)[axis_order], rshape)
    order = np.argsort(lin, kind="mergesort")
    lin = lin[order]
    row, col = lin // C, lin % C
    indptr = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=R), out=indptr[1:])
    return np.asarray(data)[order], col.astype(np.int64), indptr, compressed_axes


def uncompress_dimension(indptr):
    """reference _compressed/convert.py:82-87"""
    return np.repeat(np.arange(len(indptr) - 1, dtype=np.int64), np.diff(indptr))


def elemwise_zero_fill(op, ka, va, kb, vb, dtype=None):
    """Same-shape, zero-fill binary elementwise on canonical (sorted, unique) linear keys:
    the net effect of `_Elemwise.get_result` (reference _umath.py:457-503, masks (T,T),(T,F),
    (F,T)) + final sort, restated as a union merge.  `op` is a NumPy ufunc.  Results bit-equal
    to op(0,0) are dropped (reference :627-633)."""
    ka = np.asarray(ka, dtype=np.int64)
    kb = np.asarray(kb, dtype=np.int64)
    keys = np.union1d(ka, kb)
    xa = np.zeros(len(keys), dtype=va.dtype)
    xb = np.zeros(len(keys), dtype=vb.dtype)
    xa[np.searchsorted(keys, ka)] = va
    xb[np.searchsorted(keys, kb)] = vb
    with np.errstate(all="ignore"):
        res = op(xa, xb) if dtype is None else op(xa, xb).astype(dtype)
        fill = op(np.zeros((), va.dtype), np.zeros((), vb.dtype)).astype(res.dtype)
    ina = np.isin(keys, ka)
    inb = np.isin(keys, kb)
    # positions present in neither operand cannot occur (keys is the union)
    keep = ~_bits_equal(res, fill)
    return keys[keep], res[keep], fill, (ina, inb)


def _bits_equal(x, fill):
    x = np.ascontiguousarray(x)
    f = np.broadcast_to(np.asarray(fill, dtype=x.dtype), x.shape).copy()
    if x.dtype.kind in "fc":
        w = f"u{x.dtype.itemsize}" if x.dtype.kind == "f" else None
        if w:
            return x.view(w) == f.view(w)
        return (x.real.copy().view(f"u{x.dtype.itemsize // 2}") == f.real.copy().view(f"u{x.dtype.itemsize // 2}")) & \
               (x.imag.copy().view(f"u{x.dtype.itemsize // 2}") == f.imag.copy().view(f"u{x.dtype.itemsize // 2}"))
    return x == f


def grouped_reduce(data, groups, ufunc):
    """`_calc_counts_invidx` + `_grouped_reduce` (reference _coo/core.py:1601-1661):
    groups is a non-decreasing array; returns (values, heads, counts) with
    values = ufunc.reduceat(data, heads) (sequential left-to-right inside a segment)."""
    groups = np.asarray(groups)
    if len(groups) == 0:
        return data[:0], np.zeros(0, np.int64), np.zeros(0, np.int64)
    heads = np.flatnonzero(np.concatenate(([True], groups[1:] != groups[:-1])))
    counts = np.diff(np.concatenate((heads, [len(groups)])))
    return ufunc.reduceat(data, heads), heads, counts
