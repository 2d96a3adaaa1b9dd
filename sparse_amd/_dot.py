"""tensordot / matmul / dot and the `_dot` dispatch table of the hip backend.

Host logic only: axes normalisation, transpose+reshape to 2-D, dispatch on operand
type/`compressed_axes`/`return_type`, reshape back — the structure of the reference's
`sparse/numba_backend/_common.py:95-503`, with every jitted kernel replaced by a C-ABI call
(`_kernels.py`).  Dense operands may be NumPy arrays (copied to HBM, result copied back) or
torch tensors already on the device (zero-copy, result stays on the device).
"""
import builtins
import warnings

import numpy as np
import torch

from . import _device as dev
from . import _ffi
from . import _kernels as K
from . import _settings
from ._sparse_array import SparseArray
from ._utils import check_zero_fill_value, prod


def _is_dense(x):
    return isinstance(x, (np.ndarray, torch.Tensor))


def _is_scipy_sparse(x):
    return hasattr(x, "tocsr") and type(x).__module__.startswith("scipy.sparse")


class _DenseIO:
    """Remembers whether dense results go back to the host (NumPy in -> NumPy out)."""

    def __init__(self, *operands):
        self.numpy_out = builtins.any(isinstance(o, np.ndarray) for o in operands) or not builtins.any(
            isinstance(o, torch.Tensor) for o in operands)
        self.device = None
        for o in operands:
            d = getattr(o, "device", None)
            if isinstance(d, torch.device) and d.type == "cuda":
                self.device = d
                break

    def to_dev(self, x):
        return dev.to_device(x, self.device)

    def out(self, t):
        if isinstance(t, torch.Tensor) and self.numpy_out:
            return dev.to_numpy(t)
        return t


def tensordot_plan(a_shape, b_shape, axes=2):
    """Pure shape logic of tensordot (reference _common.py:133-198).

    The axes normalisation below is NumPy's own `numpy.tensordot` algorithm, as the reference notes at `_common.py:125`
    (NumPy is BSD-3-Clause, Copyright (c) 2005-2025 NumPy Developers); it is restated here because the shapes, the order
    of the transposed axes and the `ValueError("shape-mismatch for sum")` must be the same for the product to be a drop-in.

    Returns (newaxes_a, newshape_a, newaxes_b, newshape_b, olda, oldb) or raises the same
    ValueErrors as the reference.
    """
    try:
        iter(axes)
    except TypeError:
        axes_a = list(range(-axes, 0))
        axes_b = list(range(axes))
    else:
        axes_a, axes_b = axes
    try:
        na = len(axes_a)
        axes_a = list(axes_a)
    except TypeError:
        axes_a = [axes_a]
        na = 1
    try:
        nb = len(axes_b)
        axes_b = list(axes_b)
    except TypeError:
        axes_b = [axes_b]
        nb = 1
    nda, ndb = len(a_shape), len(b_shape)
    if nda == 0 or ndb == 0:
        if axes_a == [] and axes_b == []:
            return None
        raise ValueError(f"Input {int(nda != 0)} operand does not have enough dimensions")
    equal = na == nb
    if equal:
        for k in range(na):
            if a_shape[axes_a[k]] != b_shape[axes_b[k]]:
                equal = False
                break
            if axes_a[k] < 0:
                axes_a[k] += nda
            if axes_b[k] < 0:
                axes_b[k] += ndb
    if not equal:
        raise ValueError("shape-mismatch for sum")
    keep_a = [k for k in range(nda) if k not in axes_a]
    keep_b = [k for k in range(ndb) if k not in axes_b]
    n2a = prod(a_shape[ax] for ax in axes_a)
    n2b = prod(b_shape[ax] for ax in axes_b)
    return (keep_a + axes_a, (-1, n2a), axes_b + keep_b, (n2b, -1),
            [a_shape[ax] for ax in keep_a], [b_shape[ax] for ax in keep_b])


COO_T_CSR_MAX_NNZ = 1 << 27     # `dense @ coo`: the transposed CSR form of the COO operand is kept up to this size (see `_dot`)
TDOT_VIEW_MAX_NNZ = 1 << 18     # sparse operands up to this size keep their transposed + reshaped 2-D forms (see `_permute_reshape`)


def _permute_reshape(x, axes, shape):
    if isinstance(x, torch.Tensor):
        return x.permute(*axes).reshape(*shape)
    if isinstance(x, np.ndarray):
        return x.transpose(axes).reshape(shape)
    # A SMALL sparse operand keeps the 2-D form a `tensordot` made of it, per (axes, shape), with its other derived layouts
    # (dropped when a stored buffer changes): repeated contractions of one operand - the reference's own tensordot benchmark,
    # benchmarks/test_tensordot.py:52-68 - then skip the key permutation, its sort and the re-linearisation (6 of ~20 launches
    # at those sizes), and the 2-D form keeps ITS row pointers / CSR view across calls.  The reference memoises the same
    # conversions when asked to (`COO(cache=True)`, _coo/core.py:317-338); large operands are not kept (a second copy of them).
    # (round 6: a contraction over the TRAILING axes in their own order permutes nothing - the 2-D form is a reshape that shares the
    # operand's keys and values -, so it is kept whatever the size: a 1024^3 operand of 3 x 10^6 elements contracted with a
    # 1024 x 512 matrix rebuilt its row pointers and its block stream at every call, 0.9 ms per call for a 0.5 ms product)
    shares = list(axes) == list(range(len(axes)))
    if not hasattr(x, "__dict__") or (x.nnz > TDOT_VIEW_MAX_NNZ and not shares):
        return x.transpose(axes).reshape(shape)
    _validate_derived(x)
    views = x.__dict__.setdefault("_tdot_views", {})
    key = (tuple(axes), tuple(shape))
    v = views.get(key)
    if v is None:
        if len(views) >= 4:
            views.clear()
        v = views[key] = x.transpose(axes).reshape(shape)
    return v


TENSORDOT_WIDE_K = 1 << 22


def tensordot(a, b, axes=2, *, return_type=None):
    """Equivalent of `numpy.tensordot` for sparse/dense operand pairs
    (reference _common.py:95-215)."""
    from ._coo import COO
    from ._gcxs import GCXS

    check_zero_fill_value(a, b)
    if _is_scipy_sparse(a):
        a = GCXS.from_scipy_sparse(a)
    if _is_scipy_sparse(b):
        b = GCXS.from_scipy_sparse(b)
    plan = tensordot_plan(tuple(a.shape), tuple(b.shape), axes)
    if plan is None:  # both 0-d contraction-free: scalar product
        if a.ndim == 0 and isinstance(a, SparseArray):
            a = a.todense()
        if b.ndim == 0 and isinstance(b, SparseArray):
            b = b.todense()
        return a * b
    newaxes_a, newshape_a, newaxes_b, newshape_b, olda, oldb = plan
    if builtins.any(d == 0 for d in (*newshape_a, *newshape_b)):
        dt = np.result_type(dev.np_dtype(a.dtype) if _is_dense(a) else a.dtype,
                            dev.np_dtype(b.dtype) if _is_dense(b) else b.dtype)
        if _is_dense(a) or _is_dense(b):
            io = _DenseIO(a, b)
            if io.numpy_out:
                return np.zeros(tuple(olda + oldb), dtype=dt)
            return torch.zeros(tuple(olda + oldb), dtype=dev.torch_dtype(dt), device=io.device)
        sp = a if isinstance(a, SparseArray) else b
        return COO(np.empty((len(olda) + len(oldb), 0), dtype=np.int64), data=np.empty(0, dtype=dt),
                   shape=tuple(olda + oldb), device=sp.device)
    if isinstance(a, SparseArray) and isinstance(b, SparseArray) and return_type is None and (not olda or not oldb) \
            and int(newshape_a[1]) >= TENSORDOT_WIDE_K and a.ndim + b.ndim <= 52:
        # a sparse operand that keeps no axis is, as a matrix, one row (or column) over the product of the contracted extents:
        # 10^9 row pointers for `tensordot(x, y, axes=2)` of two 10^5 x 10^4 arrays (929 ms).  The aligned multiply + sum of
        # einsum's general route does the same contraction in 0.6 ms (late round 6, tools/r06/einsum_sweep.py).
        import string

        from ._einsum import einsum

        na = len(olda)
        ca, cb = list(newaxes_a[na:]), list(newaxes_b[:len(newaxes_b) - len(oldb)])
        la = list(string.ascii_letters[:a.ndim])
        lb, nxt = [None] * b.ndim, a.ndim
        for x_ax, y_ax in zip(ca, cb):
            lb[y_ax] = la[x_ax]
        for j in range(b.ndim):
            if lb[j] is None:
                lb[j] = string.ascii_letters[nxt]
                nxt += 1
        out = [la[i] for i in newaxes_a[:na]] + [lb[j] for j in newaxes_b[len(cb):]]
        res = einsum(f"{''.join(la)},{''.join(lb)}->{''.join(out)}", a, b)
        if res.ndim and (isinstance(a, GCXS) or isinstance(b, GCXS)) and not isinstance(res, GCXS):
            res = res.asformat("gcxs")
        return res
    at = _permute_reshape(a, newaxes_a, newshape_a)
    bt = _permute_reshape(b, newaxes_b, newshape_b)
    res = _dot(at, bt, return_type)
    return res.reshape(tuple(olda + oldb))


def check_class_nan(x, deferred=False):
    """True if any stored value / dense element is NaN (reference _common.py:72-92, the full
    pass `matmul` makes before multiplying, :245).  `deferred=True` returns either a bool (known without a scan) or a
    zero-argument callable that yields the verdict of a scan already launched on the stream."""
    from ._coo import COO
    from ._gcxs import GCXS

    if isinstance(x, (COO, GCXS)):
        data = x.data
    elif isinstance(x, np.ndarray):
        return bool(np.isnan(np.min(x))) if x.size and x.dtype.kind in "fc" else False
    elif isinstance(x, torch.Tensor):
        data = x
    else:
        raise ValueError(f"Unsupported type {type(x)}")
    if data.numel() == 0 or not (data.is_floating_point() or data.is_complex()):
        return False
    if isinstance(x, (COO, GCXS)):
        # the stored values of an array do not change between products: remember the verdict per (buffer, version)
        key = (data.data_ptr(), int(data.numel()), int(data._version))
        memo = getattr(x, "_nan_memo", None)
        if memo is not None and memo[0] == key:
            return memo[1]

        def remember(res):
            try:
                x._nan_memo = (key, res)
            except AttributeError:
                pass
            return res

        if deferred:
            probe = K.has_nan_async(data)
            return probe if isinstance(probe, bool) else _Verdict(probe, remember)
        return remember(K.has_nan(data))
    if deferred:
        probe = K.has_nan_async(data)
        return probe if isinstance(probe, bool) else _Verdict(probe, None)
    return K.has_nan(data)


class _Verdict:
    """zero-argument callable yielding the verdict of a NaN scan in flight; `ready()` tells whether it would block"""

    def __init__(self, probe, remember):
        self.probe, self.remember = probe, remember

    def ready(self):
        return self.probe.ready()

    def __call__(self):
        res = self.probe.result()
        return self.remember(res) if self.remember is not None else res


_NO_PLANS = {}


class _SpmmPlan:
    """`sparse @ dense` through the cache-less kernel (`spamd_spmm_csr`) for ONE sparse operand and one (dtype, width) of
    dense device operands, with everything that does not depend on the dense operand's contents done once: the argument
    contracts of `matmul` / `dot` (zero fill value, dimensions), the dispatch of `_dot`, the CSR triplet (index widths
    unified, contiguous), dtype codes and raw pointers.  A call then costs the validity checks below, the NaN scan of B
    (A's verdict is memoised per buffer version), `torch.empty` and one C-ABI call.  The reference's own benchmark sizes
    (benchmarks/test_benchmark_coo.py:144-176: 40-1000 stored elements) are bound by exactly this host work: 47 -> ~25 us
    per product.  Registered by `_gcxs_times_dense` after a product that took this route; dropped with the array's other
    derived layouts; declines (returns None: the general path runs) whenever anything it assumed may have changed."""

    __slots__ = ("src", "fill", "settings", "dev", "dev_index", "bdtype", "K", "N", "M", "dtr", "args", "keep", "nan_key")

    def __init__(self, a, bt, out_shape, triplet):
        data, indices, indptr = triplet
        d = a.__dict__
        self.src = tuple((name, t, int(t._version)) for name, t in ((n, d.get(n)) for n in ("data", "indices", "indptr", "_coords", "_keys"))
                         if isinstance(t, torch.Tensor))
        self.fill = d.get("fill_value")
        self.settings = (_settings.NAN_CHECK, _settings.NAN_WARNING, _settings.EXACT_MULADD, _settings.TILED_SPMM)
        self.dev, self.bdtype = bt.device, bt.dtype
        self.dev_index = bt.device.index if bt.device.index is not None else torch.cuda.current_device()
        self.M, self.N, self.K = int(out_shape[0]), int(out_shape[1]), int(bt.shape[0])
        self.dtr = dev.torch_dtype(K.dot_dtype(data.dtype, bt.dtype))
        if data.dtype != self.dtr or bt.dtype != self.dtr:
            raise TypeError("mixed value types take the general path")
        data, indices, indptr = data.contiguous(), indices.contiguous(), indptr.contiguous()
        if indices.dtype != indptr.dtype:
            raise TypeError("mixed index widths take the general path")
        self.keep = (data, indices, indptr)
        self.args = (dev.code_of(self.dtr), dev.code_of(indices.dtype), self.M, self.K, self.N, dev.ptr(data), dev.ptr(indices), dev.ptr(indptr))
        own = a.data      # (the verdict memo is kept on the array's OWN values: a csc operand's CSR twin shares them, permuted)
        self.nan_key = (own.data_ptr(), int(own.numel()), int(own._version))

    def run(self, a, b):
        d = a.__dict__
        for name, t, v in self.src:
            if d.get(name) is not t or t._version != v:
                return None
        if (b.dtype is not self.bdtype or b.shape[0] != self.K or b.device != self.dev or not b.is_contiguous()
                or d.get("fill_value") is not self.fill or b.data_ptr() % 16
                or (_settings.NAN_CHECK, _settings.NAN_WARNING, _settings.EXACT_MULADD, _settings.TILED_SPMM) != self.settings):
            return None
        probe = nan_a = False
        if self.settings[0] and self.dtr.is_floating_point:
            memo = d.get("_nan_memo")
            if memo is None or memo[0] != self.nan_key:
                return None           # (A's verdict not known yet for this buffer version: the general path scans it)
            nan_a = memo[1]
            _drain_prepared()
            probe = K.NanProbe(b)
        out = torch.empty((self.M, self.N), dtype=self.dtr, device=self.dev)
        _ffi.CALLS += 1
        rc = _SPMM_CSR(*self.args, b.data_ptr(), self.N, out.data_ptr(), self.N, _ffi.EXACT_MULADD if self.settings[2] else 0,
                       dev._raw_stream(self.dev_index))
        if rc:
            if probe is not False:
                probe.discard()
            _ffi.check(rc, "spamd_spmm_csr")
        if probe is not False or nan_a:
            if self.settings[1] == "deferred":
                _PENDING_NAN.append(nan_a or _Verdict(probe, None))
                flush_warnings(block=False)
            elif nan_a or probe.result():
                warnings.warn("Nan will not be propagated in matrix multiplication", RuntimeWarning, stacklevel=2)
        return out


def _spmm_csr_entry():
    global _SPMM_CSR
    _SPMM_CSR = _ffi.lib().spamd_spmm_csr
    return _SPMM_CSR


def _SPMM_CSR(*args):     # (bound to the library's entry point at its first use)
    return _spmm_csr_entry()(*args)


def _register_plan(a, bt, out_shape, triplet):
    """remember the row-group route of `a @ bt` (see `_SpmmPlan`); silently skipped for anything the plan does not cover"""
    if dev._raw_stream is None or not (bt.is_cuda and bt.is_contiguous()) or bt.numel() == 0 or not hasattr(a, "__dict__"):
        return
    try:
        plan = _SpmmPlan(a, bt, out_shape, triplet)
    except TypeError:
        return
    plans = a.__dict__.setdefault("_mm_plans", {})
    if len(plans) >= 8:
        plans.clear()
    plans[(bt.dtype, int(bt.shape[1]))] = plan


def matmul(a, b):
    """Equivalent of `numpy.matmul` (reference _common.py:218-293)."""
    if type(b) is torch.Tensor:
        # a product this array has made before with a dense operand of this type and width (small products are bound by
        # the host: bench_small.py): everything that depends on `a` alone was checked and converted then (`_SpmmPlan`)
        plans = getattr(a, "__dict__", _NO_PLANS).get("_mm_plans")
        if plans is not None and b.dim() == 2:
            plan = plans.get((b.dtype, b.shape[1]))
            if plan is not None:
                res = plan.run(a, b)
                if res is not None:
                    return res
    check_zero_fill_value(a, b)
    if not hasattr(a, "ndim") or not hasattr(b, "ndim"):
        raise TypeError(f"Cannot perform dot product on types {type(a)}, {type(b)}")
    if not _settings.NAN_CHECK:
        return _matmul(a, b)
    # the reference scans both operands before multiplying (_common.py:245-246) only to WARN; here the scans are
    # launched, the product is queued behind them, and the verdicts are read afterwards from pinned host memory (the
    # host waits for the scan kernels only, never for the product)
    _drain_prepared()
    probes = [check_class_nan(a, deferred=True)]
    try:
        probes.append(check_class_nan(b, deferred=True))
        res = _matmul(a, b)
    except BaseException:
        for p in probes:    # the product raised: hand the scans' verdict slots back unread
            if isinstance(p, _Verdict):
                p.probe.discard()
        raise
    if _settings.NAN_WARNING == "deferred":
        _PENDING_NAN.extend(p for p in probes if p is not False)
        flush_warnings(block=False)
        return res
    if builtins.any(p if isinstance(p, bool) else p() for p in probes):
        warnings.warn("Nan will not be propagated in matrix multiplication", RuntimeWarning, stacklevel=1)
    return res


_PENDING_NAN = []   # verdicts not read yet (`_settings.NAN_WARNING == "deferred"`): True, or a callable of a scan in flight


def flush_warnings(block=True):
    """Raise the NaN warnings of products whose scans have finished (`block=True`: wait for all of them)."""
    found, keep = False, []
    for p in _PENDING_NAN:
        if isinstance(p, bool):
            found = found or p
        elif block or getattr(p, "ready", lambda: True)():
            found = bool(p()) or found
        else:
            keep.append(p)
    _PENDING_NAN[:] = keep
    if found:
        warnings.warn("Nan will not be propagated in matrix multiplication", RuntimeWarning, stacklevel=2)


def _matmul(a, b):
    if b.ndim <= 2:
        return dot(a, b)
    if a.ndim <= 2:
        res = dot(a, b)
        axes = list(range(res.ndim))
        axes.insert(-1, axes.pop(0))
        return res.permute(*axes) if isinstance(res, torch.Tensor) else res.transpose(axes)
    if a.ndim <= b.ndim and prod(a.shape[:-1]) == 1:
        res = dot(a.reshape(-1), b)
        shape = list(res.shape)
        shape.insert(-1, 1)
        return res.reshape(shape)
    if b.ndim <= a.ndim and prod(b.shape[:-2]) == 1:
        return dot(a, b.reshape(tuple(b.shape[-2:])))
    if a.ndim < b.ndim:
        a = a[(None,) * (b.ndim - a.ndim)]
    if a.ndim > b.ndim:
        b = b[(None,) * (a.ndim - b.ndim)]
    for i, j in zip(a.shape[:-2], b.shape[:-2]):
        if i != 1 and j != 1 and i != j:
            raise ValueError("shapes of a and b are not broadcastable")
    from ._batched import matmul_batched, matmul_blockdiag, matmul_broadcast

    if prod(a.shape) == 0 or prod(b.shape) == 0:
        return matmul_batched(a, b)     # empty batches / matrices: the reference's recursion handles every corner
    if isinstance(a, SparseArray) and tuple(a.shape[:-2]) == tuple(b.shape[:-2]):
        return matmul_blockdiag(a, b)   # the whole batch as one block-diagonal product
    return matmul_broadcast(a, b)       # broadcast leading axes (and dense @ sparse): still one product


def dot(a, b):
    """Equivalent of `numpy.dot` (reference _common.py:296-336)."""
    check_zero_fill_value(a, b)
    if not hasattr(a, "ndim") or not hasattr(b, "ndim"):
        raise TypeError(f"Cannot perform dot product on types {type(a)}, {type(b)}")
    if a.ndim == 1 and b.ndim == 1:
        from ._coo import as_coo

        if isinstance(a, SparseArray):
            a = as_coo(a)
        if isinstance(b, SparseArray):
            b = as_coo(b)
        return (a * b).sum()
    if a.ndim == 2 and b.ndim == 2 and not _is_scipy_sparse(a) and not _is_scipy_sparse(b) \
            and a.shape[1] == b.shape[0] and a.shape[0] and a.shape[1] and b.shape[1]:
        # matrix x matrix: tensordot's axis normalisation, transposes and reshapes are all the identity here (the common
        # case, and the one whose host cost bounds loops of short products)
        return _dot(a, b, None)
    a_axis, b_axis = -1, -2
    if b.ndim == 1:
        b_axis = -1
    return tensordot(a, b, axes=(a_axis, b_axis))


def _return_kind(return_type):
    """Normalise `return_type` to one of None / "ndarray" / "coo" / "gcxs"."""
    from ._coo import COO
    from ._gcxs import GCXS

    if return_type is None:
        return None
    if return_type is np.ndarray or return_type is torch.Tensor:
        return "ndarray"
    if return_type is COO:
        return "coo"
    if return_type is GCXS:
        return "gcxs"
    # the reference's own classes (drop-in callers may pass sparse.COO / sparse.GCXS)
    name = getattr(return_type, "__name__", "")
    if name in ("COO", "GCXS"):
        return name.lower()
    raise TypeError(f"unsupported return_type {return_type!r}")


def _tiled_dtype(data, bt):
    """value type of the tiled kernel for this product (the reference's result dtype rule `_dot_dtype`,
    _common.py:635-636, restricted to what the kernel covers): float32 x float32, or float64 with float32/64."""
    if data.dtype == torch.float32 and bt.dtype == torch.float32:
        return torch.float32
    if {data.dtype, bt.dtype} <= {torch.float32, torch.float64}:
        return torch.float64
    if data.dtype == torch.int32 and bt.dtype == torch.int32:
        return torch.int32      # (round 4: exact wrap-around products, the executor's int32 variant)
    return None


LDSB_MAX_K = 639      # (K + 1) rows of 256 bytes within the 160 KB of LDS: `spamd_spmm_csr_ldsb_fits` (575 / 144 KB until late round 4)


def _tiled_eligible(data, bt, out_shape, Kd):
    """The inspector/executor kernel covers float32 and float64 products whose B is (padded to) whole 128- / 64-column
    panels; from N = 5 (float32: 8 until late round 4) on the padded product beats the row-group kernel (round 3, config-2 operand:
    fp32 N = 8 / 16 / 32: 0.80 vs 1.14 / 1.17 / 1.18 ms, fp64 N = 5 / 8 / 16: 1.02 vs 1.17 / 1.33 / 1.32 ms;
    `NARROW=1 tools/rowgroup_shapes.py`; late round 4, tools/r04/narrow_n.py: fp32 N = 5 / 6 / 7 0.84 / 0.77 / 0.83 vs 0.86 /
    1.08 / 0.83 ms - odd widths paid a slice of the padded result; with the single-column store of the straddling lane
    0.80 / 0.78 / 0.78 ms, so fp32 starts at N = 5 as well; results of at most 4 columns have the row-vector kernel).  Thresholds
    measured on MI355X (tools/tiled_crossover.py, tools/r04/m_crossover.py): enough rows for the width (`_tiled_min_rows`)
    and enough stored elements per 32 x 128 cells for the width (`_tiled_min_density`: 5-12).  The inspector costs about one row-group product, so
    a single product breaks even and every further one is 2-3x faster."""
    M, N = out_shape
    if _settings.TILED_SPMM == "never" or bt.dim() != 2:
        return False
    dt = _tiled_dtype(data, bt)
    if dt is None or N < 5:   # narrower results: the row-vector kernel
        return False
    if (Kd + 512) * N * bt.element_size() >= (1 << 32):   # the executor walks B with 32-bit byte offsets (buffer-form tile DMA)
        return False
    if Kd <= LDSB_MAX_K and N * bt.element_size() >= 128:
        # a short contracted axis: `spamd_spmm_csr` keeps a column panel of B in LDS by itself (spmm_ldsb.hip) - as fast as
        # the executor on these shapes (config 3: 0.13-0.15 ms against 0.143 ms) without an inspector or a second copy of A
        return False
    per_list = int(data.numel()) * 4096 / max(M * Kd, 1)
    return M >= _tiled_min_rows(N * dt.itemsize) and per_list >= _tiled_min_density(N * dt.itemsize, Kd * N * bt.element_size())


def _tiled_min_density(row_bytes, b_bytes):
    """Fewest stored elements per 4096 cells (a 32 x 128 patch) at which the executor wins.  Its time barely depends on the
    density below ~0.5 % (it walks every list, empty or not), the row-group kernel's is linear in it and in the width.
    Round 4, M = 262144, K = 10^4 (tools/r04/density_crossover.py; executor / row-group ms at 4.1, 8.2, 12.3 per 4096):
      fp32 N = 128: 0.156 / 0.125, 0.157 / 0.172, 0.162 / 0.236;      fp32 N = 512: 0.610 / 0.684, 0.609 / 1.26, 0.630 / 1.82;
      fp64 N = 128: 0.304 / 0.264, 0.310 / 0.479, 0.314 / 0.694;      fp64 N = 512: 1.21 / 1.52, 1.23 / 2.83, 1.25 / 4.19;
    K = 10^5 (B of 51-410 MB: the row-group kernel's gathers miss the caches) crosses at ~3.4 for every width.
    Rounds 1-3 asked for 12 (6 with a large B) whatever the width."""
    if row_bytes < 512:
        return 6 if b_bytes >= (16 << 20) else 12
    panels = -(-row_bytes // 512)
    need = 9 if panels == 1 else (6 if panels == 2 else 5)
    return min(need, 5) if b_bytes >= (16 << 20) else need


def _tiled_min_rows(row_bytes):
    """Fewest rows at which the executor beats the row-group kernel, by the width of a result row.  The executor's time
    has a floor (every workgroup walks all of K: 0.108 ms at K = 10^4) but does not grow with the number of column
    panels until the chip is full, while the row-group kernel's grows with M x N and doubles again for 8-byte values.
    Round 4, 1 % density, K = 10^4 (tools/r04/m_crossover.py; executor / row-group ms):
      fp32 N = 128: M = 40960 0.109 / 0.106, 50000 0.112 / 0.129, 65536 0.113 / 0.167;
      fp32 N = 512: M = 4096 0.109 / 0.068, 8192 0.109 / 0.165, 16384 0.113 / 0.342, 32768 0.125 / 0.699;
      fp64 N = 128: M = 8192 0.115 / 0.064, 16384 0.119 / 0.130, 32768 0.120 / 0.264;
      fp64 N = 512: M = 4096 0.118 / 0.171, 16384 0.149 / 0.788, 65536 0.542 / 3.24.
    (Rounds 1-3 asked for 65536 rows whatever the width - 117 workgroups of 560 rows - and so left the 3-6x of the wide
    and float64 cases unused.)  Partial panels keep the old bound: the row-group kernel is at its best on narrow rows."""
    if row_bytes < 512:
        return 65536
    panels = -(-row_bytes // 512)
    return 45056 if panels == 1 else max(4096, 40960 // panels)


DERIVED_CACHES = ("_mm_plans", "_keys2d", "_tdot_views", "_csr_of_t", "_csr_view", "_csr_twin", "_tiled_layouts", "_spmm_uses", "_nan_memo", "_derived_stamp", "_sddmm_plan", "_t_view", "_coo_view", "_hot_split")


def drop_derived(a):
    """Forget every layout derived from the stored arrays of `a` (CSR view / twin, tiled block streams, NaN verdict).
    Called whenever the container's buffers are replaced (`_make_shallow_copy_of`, the `out=` path of ufuncs)."""
    for name in DERIVED_CACHES:
        a.__dict__.pop(name, None)


def _stamp(a):
    """Identity + version of every stored buffer the derived layouts are built from: in-place writes (`a.data *= 2`,
    `a.data[i] = x`) bump torch's version counter, a replaced buffer changes the pointer."""
    bufs = [a.data]
    if hasattr(a, "indices"):
        bufs += [a.indices, a.indptr]
    else:
        c = a.__dict__.get("_coords")
        k = getattr(a, "_keys", None)
        bufs += [t for t in (c, k) if t is not None]
    return tuple((t.data_ptr(), int(t.numel()), int(t._version)) for t in bufs if isinstance(t, torch.Tensor))


def _validate_derived(a):
    """Drop the derived layouts of `a` if any stored buffer changed since they were built."""
    st = _stamp(a)
    if a.__dict__.get("_derived_stamp") != st:
        drop_derived(a)
        a.__dict__["_derived_stamp"] = st


def prepare_spmm(a, dtype=None, force_sort=False):
    """Build (and cache on `a`) the tiled block stream used by `a @ dense` for value type `dtype` (default: a's own
    if float32/float64); returns True if `a` now has one.  The counterpart of the reference's memoised conversions
    (`COO(cache=True)`, _coo/core.py:317-338)."""
    from ._coo import COO
    from ._gcxs import GCXS

    if not isinstance(a, (GCXS, COO)) or a.ndim != 2:
        return False
    dtype = dtype or (a.data.dtype if a.data.dtype in K.TILED_DTYPES else None)
    if dtype not in K.TILED_DTYPES:
        return False
    _validate_derived(a)
    layouts = a.__dict__.setdefault("_tiled_layouts", {})
    if dtype not in layouts:
        if not force_sort and _csc_without_twin(a) and a.nnz >= CSC_INSPECT_MIN_NNZ:
            # a csc operand (the reference's default for a tall matrix) that has no CSR twin yet: the block stream is built
            # from the CSC arrays themselves (K.csc_tiled_layout) - no conversion, no twin
            lay = K.csc_tiled_layout(a.data, a.indices, a.indptr, int(a.shape[0]), int(a.shape[1]), dtype=dtype)
            if lay is not None:
                layouts[dtype] = lay
                return True
        d, i, p = _csr_triplet(a)
        layouts[dtype] = K.csr_tiled_layout(d, i, p, int(a.shape[0]), int(a.shape[1]), dtype=dtype, force_sort=force_sort,
                                            defer_check=True)
    return True


# Late round 4 (tools/r04/csc_inspect.py; CSC inspector against CSC -> CSR + CSR inspector, ms): 4.2 x 10^4 stored elements
# 0.07 / 0.06, 1.2 x 10^5 0.09 / 0.08, 10^6 0.07 / 0.14, 8 x 10^6 0.18 / 0.36, 10^8 1.36 / 3.4 (float64: 1.5 / 5.2).
CSC_INSPECT_MIN_NNZ = 200_000


def _csc_without_twin(a):
    from ._gcxs import GCXS

    return isinstance(a, GCXS) and a.ndim == 2 and a.compressed_axes == (1,) and a.__dict__.get("_csr_twin") is None


def prepare_operand(a, b_like):
    """Everything `matmul(a, dense)` needs from `a` alone, queued now: the NaN scan of its values (the verdict is
    memoised per buffer and version, so the product's own check is then a dictionary hit) and, when the product with an
    operand like `b_like` (its dtype and column count) takes the inspector/executor kernel, the block stream.  Used by the
    sharded products to fill the time the all-gather of B is in flight; harmless (and idempotent) anywhere else."""
    from ._coo import COO
    from ._gcxs import GCXS

    if not isinstance(a, (GCXS, COO)) or a.ndim != 2 or not isinstance(b_like, torch.Tensor) or b_like.dim() != 2:
        return
    if _settings.NAN_CHECK:
        v = check_class_nan(a, deferred=True)
        if isinstance(v, _Verdict):
            _PENDING_PREP.append(v)     # read (and memoised on `a`) by the next `matmul` through `_drain_prepared`
    if isinstance(a, GCXS) or getattr(a, "_tiled_layouts", None):
        data = a.data if _csc_without_twin(a) else _csr_triplet(a)[0]
        out_shape = (int(a.shape[0]), int(b_like.shape[1]))
        if _tiled_eligible(data, b_like, out_shape, int(a.shape[1])):
            prepare_spmm(a, _tiled_dtype(data, b_like))


_PENDING_PREP = []   # NaN scans started by `prepare_operand`, not read yet


def _drain_prepared():
    """Read the verdicts of `prepare_operand`'s scans (each remembers its result on its array)."""
    while _PENDING_PREP:
        _PENDING_PREP.pop()()


def _tiled_product(a, dt, out_shape, Kd, b):
    """Executor product from `a`'s cached layout.  The one-pass inspector's "unsorted column indices" verdict is read
    behind the first product's launch (no host wait between inspector and executor); if it says unsorted — rows whose
    column indices do not ascend (reachable: `GCXS((data, indices, indptr))` takes the caller's arrays as they are) — the
    layout is rebuilt by the key-sort recipe and the product repeated.  The first, discarded product is memory-safe: the
    inspector writes zero entries for every row group that met such a row (csrc/spmm_tiled.hip, `group_bad`)."""
    try:
        lay = a._tiled_layouts[dt]
        res = K.dot_csr_ndarray_tiled(lay, out_shape, Kd, b, exact=_settings.EXACT_MULADD)
        if getattr(lay, "rebalance", False):
            # the deferred skew test (read behind this product's launch) says the rows are skewed: every later product takes a
            # balanced layout (csrc/spmm_tiled.hip `tl_map_*`); this one is correct as it is
            lay.rebalance = False
            d, i, p = _csr_triplet(a)
            a._tiled_layouts[dt] = K.csr_tiled_layout(d, i, p, int(a.shape[0]), int(a.shape[1]), dtype=dt, balance=True)
        return res
    except K.UnsortedColumns:
        del a._tiled_layouts[dt]
        prepare_spmm(a, dt, force_sort=True)
        return K.dot_csr_ndarray_tiled(a._tiled_layouts[dt], out_shape, Kd, b, exact=_settings.EXACT_MULADD)


def _csr_triplet(a):
    """(data, indices, indptr) of a 2-D GCXS compressed by rows; a csc array is re-compressed once (stable
    key sort) and the CSR twin memoised on the (immutable) array — the reference's `format="gcxs"` default is
    compressed_axes=(argmin(shape),), so tall matrices arrive as csc."""
    from ._coo import COO

    _validate_derived(a)
    if isinstance(a, COO):  # canonical 2-D COO is CSR order already: only the row pointers are missing
        view = getattr(a, "_csr_view", None)
        if view is None:
            keys = getattr(a, "_keys", None)
            it = a._index_dtype
            if a.__dict__.get("_coords") is None and keys is not None and it in (torch.int32, torch.int64) \
                    and (it == torch.int64 or max(int(a.shape[1]), a.nnz) < 2 ** 31):
                # built from linear keys (a reshape / transpose / elementwise result) and the coordinates never asked for:
                # column indices and row pointers come from the keys in one pass, the coordinate rows are not materialised
                indptr, indices = K.keys_to_csr(keys, int(a.shape[0]), int(a.shape[1]), it)
                view = (a.data, indices, indptr)
            else:
                cols = a.coords[1].contiguous()
                indptr = K.rows_to_indptr(a.coords[0], int(a.shape[0]))
                if cols.dtype == torch.int32 and a.nnz < 2 ** 31:
                    # the row pointers (R + 1 words) in the coordinates' own width: with int64 pointers every kernel that
                    # takes the triplet would first widen the nnz column indices (`_unify_index`: 0.3 ms and 0.8 GB at
                    # config 2's size, and an inspector that reads 8-byte indices: 0.89 instead of 0.51 ms)
                    indptr = indptr.to(torch.int32)
                view = (a.data, cols, indptr)
            a._csr_view = view
        return view
    if a.compressed_axes == (0,):
        return a.data, a.indices, a.indptr
    twin = getattr(a, "_csr_twin", None)
    if twin is None:
        twin = K._csc_to_csr(a.shape, a.data, a.indices, a.indptr)
        a._csr_twin = twin
    return twin


COO_TILED_FIRST_NNZ = 20_000_000
COO_TILED_FIRST_BYTES = 4_000_000_000


def _coo_first_product_tiled(nnz, row_bytes):
    """Does the FIRST eligible product of a COO operand pay for its inspector?  Results of one 512-byte column panel: from
    COO_TILED_FIRST_NNZ stored elements (round 4: 2.9 -> 1.73 ms at config 2's size; 0.54 against 0.50 ms at 1.6 x 10^7).
    Wider and float64 results, where the row-group kernel's cost per element grows with the width: from
    nnz x (bytes of a result row) = COO_TILED_FIRST_BYTES - tools/r04/coo_first_sweep.py, inspector + executor against
    the row-group kernel, ms: fp32 N = 512: 2 x 10^6 elements 0.38 / 0.52, 4 x 10^6 0.42 / 0.95, 4 x 10^7 1.94 / 8.95;
    fp64 N = 128: 2 x 10^6 0.32 / 0.22, 4 x 10^6 0.34 / 0.39, 1.6 x 10^7 0.90 / 1.45; fp64 N = 512: 4 x 10^6 0.71 / 2.15."""
    if nnz >= COO_TILED_FIRST_NNZ:
        return True
    return row_bytes > 512 and nnz * row_bytes >= COO_TILED_FIRST_BYTES


HOT_ROW_SPLIT = True
HOT_ROW_MIN = 4096            # stored elements from which a row is "hot" (one wave walks a row at ~27-120 ns per element)
HOT_ROW_MIN_STREAM = 32768    # ... for the widths the stream kernel takes: it deals a row of 10^4 elements to its waves by
                              # itself (0.05 ms against 0.13-0.18 in two parts), a row of 10^5 costs it 0.37 ms against 0.13
HOT_ROW_MIN_NNZ = 1 << 14     # operands with fewer stored elements are not probed (two small kernels and a read, ~40 us once
                              # per operand; a row of 49 000 elements in a 1 x K operand: 5.6 ms at 16 columns unsplit)
HOT_ROWS_MAX = 1024           # more hot rows than this: the matrix simply has long rows, nothing is split
HOT_PIECE_GROUPS = 1024       # row groups (of 35 rows) the pieces of the hot rows should fill
HOT_COMBINE_FAN = 128         # pieces a thread of the combine kernel adds in one loop


def hot_piece_plan(p0, p1, groups=None, fan=None):
    """Host arithmetic of the hub-row split, from the hot rows' element ranges [p0[h], p1[h]) alone:
    (vptr, vfirst, g1) - `vptr` = row pointers of the piece matrix over the hot rows' elements laid end to end (pieces of one
    size, 32 .. 4096 elements and a multiple of 32, chosen so that all pieces fill `groups` row groups of 35 rows; a row's
    last piece may be shorter), `vfirst[h]` = first combine input of hot row h, `g1` = None or, when a row has more than `fan`
    pieces (a thread of the combine kernel adds its inputs one after the other: 31 250 pieces in one loop took 6.8 ms), the
    first piece of every group of at most `fan` pieces of one row - `vfirst` then counts those groups."""
    groups = groups or HOT_PIECE_GROUPS
    fan = fan or HOT_COMBINE_FAN
    lens = np.asarray(p1, dtype=np.int64) - np.asarray(p0, dtype=np.int64)
    total = int(lens.sum())
    piece = int(min(4096, max(32, total // (groups * 35) // 32 * 32)))
    vptr, vfirst, base = [0], [0], 0
    for n in lens.tolist():
        vptr.extend((base + np.minimum(np.arange(piece, n + piece, piece), n)).tolist())
        vfirst.append(len(vptr) - 1)
        base += n
    g1 = None
    if max(b - a_ for a_, b in zip(vfirst[:-1], vfirst[1:])) > fan:
        g1, f2 = [0], [0]
        for a_, b in zip(vfirst[:-1], vfirst[1:]):
            g1.extend(min(s0 + fan, b) for s0 in range(a_, b, fan))
            f2.append(len(g1) - 1)
        vfirst, g1 = f2, np.asarray(g1, dtype=np.int64)
    return np.asarray(vptr, dtype=np.int64), np.asarray(vfirst, dtype=np.int64), g1


def _hot_row_split(a, data, indices, indptr):
    """None, or what `_gcxs_times_dense` needs to multiply an operand with a few hub rows in two parts (csrc/hot_rows.hip):
    (the matrix without its hot rows, the hot rows cut into pieces that are rows of their own, first piece of every hot row,
    the hot rows' numbers).  Found once per operand - the longest row from the row pointers, one small read - and kept with
    it; both parts are GCXS arrays of their own, so their layouts are memoised like anybody's."""
    from ._gcxs import GCXS
    from ._reduce import reduce_all
    from ._umath import binary_arrays

    if "_hot_split" in a.__dict__:
        return a.__dict__["_hot_split"]
    res = None
    M = int(indptr.numel()) - 1
    nnz = int(data.numel())
    if nnz >= HOT_ROW_MIN_NNZ and M >= 1 and not a.__dict__.get("_no_hot_split") and indptr.dtype in (torch.int32, torch.int64):
        lens = binary_arrays("subtract", indptr[1:].clone(), indptr[:-1].contiguous())     # (the clone: an aligned start)
        longest = int(reduce_all(lens, "maximum")[1])
        if longest >= HOT_ROW_MIN:
            devi = data.device
            lens = lens.to(torch.int64)
            ge = binary_arrays("greater_equal", lens, torch.tensor([HOT_ROW_MIN], dtype=torch.int64, device=devi), b_scalar=True,
                               out_bool_as=torch.uint8)
            flags = K.flag_ne_bits(ge, np.uint8(0))
            offs = K.exclusive_scan(flags)
            H = int(offs[-1])
            if 1 <= H <= HOT_ROWS_MAX:
                rows = K.compact(torch.arange(M, dtype=torch.int64, device=devi), flags, offs, H)
                p64 = indptr.to(torch.int64)
                p0 = K.gather(p64, rows).cpu().numpy()
                hl = K.gather(lens, rows)
                p1 = p0 + hl.cpu().numpy()
                # the matrix without the hot rows: the stretches between them, and the pointers less what went before
                cuts = np.concatenate(([0], np.stack([p0, p1], axis=1).reshape(-1), [nnz]))
                keep = [(int(cuts[i]), int(cuts[i + 1])) for i in range(0, len(cuts), 2) if cuts[i + 1] > cuts[i]]
                cat = lambda t, spans: (torch.cat([t[s:e] for s, e in spans]) if spans else t[:0].clone())
                d = torch.zeros(M + 1, dtype=torch.int64, device=devi)
                d[rows] = hl
                removed = K.exclusive_scan(d)                               # [i] = hot elements in rows before i
                light = GCXS((cat(data, keep), cat(indices, keep), (p64 - removed).to(indptr.dtype)), shape=(M, int(a.shape[1])),
                             compressed_axes=(0,), fill_value=a.fill_value)
                # the hot rows in pieces: enough of them to fill the chip, each a row of a (pieces x K) matrix
                hot_spans = [(int(s0), int(e0)) for s0, e0 in zip(p0, p1)]
                vptr, vfirst, g1 = hot_piece_plan(p0, p1)
                hotv = GCXS((cat(data, hot_spans), cat(indices, hot_spans),
                             torch.tensor(vptr, device=devi).to(indptr.dtype)),
                            shape=(len(vptr) - 1, int(a.shape[1])), compressed_axes=(0,), fill_value=a.fill_value)
                light.__dict__["_no_hot_split"] = hotv.__dict__["_no_hot_split"] = True
                mid = None
                if g1 is not None:
                    mid = (torch.tensor(g1, device=devi), torch.arange(len(g1) - 1, dtype=torch.int64, device=devi))
                res = (light, hotv, torch.tensor(vfirst, device=devi), rows, mid, longest)
    a.__dict__["_hot_split"] = res
    return res


_HOT_COMBINE_TYPES = (torch.float32, torch.float64, torch.int32, torch.int64)


def _hot_product(split, bt, out_shape):
    light, hotv, vfirst, rows, mid = split[:5]
    N = int(out_shape[1])
    out = _gcxs_times_dense(light, bt, out_shape)
    part = _gcxs_times_dense(hotv, bt, (int(hotv.shape[0]), N))
    if part.dtype != out.dtype or out.dtype not in _HOT_COMBINE_TYPES or not (out.is_contiguous() and part.is_contiguous()):
        raise _ffi.HipBackendError(f"hot-row parts of types {out.dtype} / {part.dtype} cannot be combined")
    code, st = dev.code_of(out.dtype), dev.stream_ptr(out.device)
    if mid is not None:
        g1, ident = mid
        tmp = torch.empty((int(ident.numel()), N), dtype=out.dtype, device=out.device)
        _ffi.call("spamd_hot_rows_combine", code, int(ident.numel()), N, dev.ptr(part), N, dev.ptr(g1), dev.ptr(ident), dev.ptr(tmp), N, st)
        part = tmp
    _ffi.call("spamd_hot_rows_combine", code, int(rows.numel()), N, dev.ptr(part), N, dev.ptr(vfirst), dev.ptr(rows), dev.ptr(out), N, st)
    return out


def _gcxs_times_dense(a, bt, out_shape):
    """GCXS or canonical 2-D COO times dense."""
    from ._coo import COO

    _validate_derived(a)
    Kd = int(a.shape[1])
    direct = _csc_without_twin(a)     # (eligibility needs the values' count and type only: no CSR twin for the executor's sake)
    data = a.data if direct else None
    if direct and not (_tiled_eligible(data, bt, out_shape, Kd) and
                       (a.nnz >= CSC_INSPECT_MIN_NNZ or (getattr(a, "_tiled_layouts", None) or {}).get(_tiled_dtype(data, bt)))):
        direct = False
    if not direct:
        data, indices, indptr = _csr_triplet(a)
        if HOT_ROW_SPLIT and not _settings.EXACT_MULADD and int(out_shape[1]) > 0 and \
                K.torch_dtype(K.dot_dtype(data.dtype, bt.dtype)) in _HOT_COMBINE_TYPES:
            split = _hot_row_split(a, data, indices, indptr)
            if split is not None and (split[5] >= HOT_ROW_MIN_STREAM or not (
                    int(out_shape[1]) <= K.STREAM_MULTI_MAX_N and K.stream_passes(
                        int(out_shape[0]), Kd, int(out_shape[1]), K.torch_dtype(K.dot_dtype(data.dtype, bt.dtype)), data, indices))):
                return _hot_product(split, bt, out_shape)
    use_tiled = eligible = _tiled_eligible(data, bt, out_shape, Kd)
    if use_tiled and not direct and out_shape[1] <= K.STREAM_MULTI_MAX_N and not (
            _settings.EXACT_MULADD and _tiled_dtype(data, bt).is_floating_point):
        # a result of 5-12 columns from CSR arrays that are there anyway: two or three passes of the stream kernel over A
        # (0.36-0.53 ms at config 2's matrix) instead of the executor's padded 128-column panel (0.77 ms) - and no inspector
        dt = _tiled_dtype(data, bt)
        if K.stream_passes(int(out_shape[0]), Kd, int(out_shape[1]), dt, data, indices):
            use_tiled = eligible = False
    if use_tiled and isinstance(a, COO) and not getattr(a, "_tiled_layouts", None):
        # COO operands of `tensordot` are usually temporaries (an N-D array reshaped to 2-D): for small ones the inspector
        # only pays when the array is multiplied again (config 3, 1.3 x 10^6 elements: 0.33 ms with a fresh layout per call
        # against 0.14), so such a COO gets its block stream at its SECOND eligible product.  From COO_TILED_FIRST_NNZ stored
        # elements on the first product already takes it: the fixed part of a fresh layout (allocations, three launches,
        # the verdict read-back: ~0.15 ms) is then below what the executor saves over the cache-less kernel (config 2's
        # size: inspector 0.57 + executor 0.85 ms against 2.8 ms).
        a._spmm_uses = getattr(a, "_spmm_uses", 0) + 1
        use_tiled = a._spmm_uses >= 2 or _coo_first_product_tiled(int(data.numel()), out_shape[1] * _tiled_dtype(data, bt).itemsize)
    if use_tiled:
        # the inspector costs about one product (1.25 ms at config 2 against 0.85 ms per tiled and 2.8 ms per
        # row-group product), so it runs at the first eligible product and is cached on the array
        dt = _tiled_dtype(data, bt)
        prepare_spmm(a, dt)
        M, N = out_shape
        panel = 64 if dt == torch.float64 else 128
        bt = bt.to(dt)
        if N % panel:
            # a workgroup covers whole 512-byte column panels: B (small) is zero-padded to the next panel; the result is
            # NOT - the last panel stores its leading columns only (float32: a lane holds two columns; the lane that
            # straddles the end of an odd-width row stores its first one alone - late round 4, a padded result + slice before)
            npad = -(-N // panel) * panel
            bp = torch.zeros((Kd, npad), dtype=dt, device=bt.device)
            bp[:, :N] = bt
            return _tiled_product(a, dt, (M, N), Kd, bp)
        return _tiled_product(a, dt, out_shape, Kd, bt)
    res = K.dot_csr_ndarray(out_shape, data, indices, indptr, bt, exact=_settings.EXACT_MULADD)
    if a.ndim == 2 and not eligible:     # (an eligible COO that waits for its second product must keep counting its products)
        _register_plan(a, bt, out_shape, (data, indices, indptr))
    return res



def _dot(a, b, return_type=None):
    """2-D x 2-D product: the dispatch table of reference `_common.py:339-503` (Appendix B of
    SURVEY.md), one C-ABI kernel per row of the table."""
    from ._coo import COO
    from ._gcxs import GCXS

    rk = _return_kind(return_type)
    out_shape = (int(a.shape[0]), int(b.shape[1]))
    io = _DenseIO(a, b)

    if isinstance(a, SparseArray) and isinstance(b, SparseArray) and (
            isinstance(a, GCXS) or isinstance(b, GCXS)):
        a = a.asformat("gcxs")
        b = b.asformat("gcxs", compressed_axes=a.compressed_axes)

    if isinstance(a, GCXS) and isinstance(b, GCXS):
        if a.nbytes > b.nbytes:
            b = b.change_compressed_axes(a.compressed_axes)
        else:
            a = a.change_compressed_axes(b.compressed_axes)
        if a.compressed_axes == (0,):  # csr @ csr
            ca = (0,)
            data, indices, indptr = K.dot_csr_csr(out_shape, a.data, b.data, a.indices, b.indices,
                                                  a.indptr, b.indptr)
        else:  # csc @ csc:  a @ b = (b.T @ a.T).T, the transposes being free
            ca = (1,)
            data, indices, indptr = K.dot_csr_csr(out_shape[::-1], b.data, a.data, b.indices, a.indices,
                                                  b.indptr, a.indptr)
        out = GCXS((data, indices, indptr), shape=out_shape, compressed_axes=ca, prune=True)
        if rk == "ndarray":
            return io.out(out.todense_device())
        if rk == "coo":
            return out.tocoo()
        return out

    if isinstance(a, GCXS) and _is_dense(b):
        bt = io.to_dev(b) if not (isinstance(b, torch.Tensor) and b.is_cuda) else b
        bt = dev.to_device(bt, a.device)
        if rk in (None, "ndarray"):
            return io.out(_gcxs_times_dense(a, bt, out_shape))
        if a.compressed_axes == (0,):  # csr @ dense, sparse result
            data, indices, indptr = K.dot_csr_ndarray_sparse(out_shape, a.data, a.indices, a.indptr, bt)
            out = GCXS((data, indices, indptr), shape=out_shape, compressed_axes=(0,), prune=True)
            return out.tocoo() if rk == "coo" else out
        data, indices, indptr = K.dot_csc_ndarray_sparse(a.shape, tuple(bt.shape), a.data, a.indices,
                                                         a.indptr, bt)
        out = GCXS((data, indices, indptr), shape=out_shape, compressed_axes=(1,), prune=True)
        return out.tocoo() if rk == "coo" else out

    if _is_dense(a) and isinstance(b, GCXS):
        # dense @ sparse == (sparse.T @ dense.T).T ; sparse.T is free for 2-D GCXS
        ad = dev.to_device(a, b.device)
        at = ad.t()
        # the transposed view shares b's buffers; it is kept on b (and dropped with b's other derived layouts) so that
        # what it caches - the CSR twin of a csc operand, the K-tiled block streams of the executor - survives the call
        bt = b.__dict__.get("_t_view")
        if bt is None or bt.data is not b.data or bt.indices is not b.indices:
            bt = b.T
            b.__dict__["_t_view"] = bt
        if rk in (None, "ndarray"):
            # same kernels and the same inspector/executor policy as sparse @ dense (A1): (b^T a^T)^T
            res = _gcxs_times_dense(bt, K.transposed_copy(ad), out_shape[::-1])
            return io.out(res.t())
        if bt.compressed_axes == (0,):
            data, indices, indptr = K.dot_csr_ndarray_sparse(out_shape[::-1], bt.data, bt.indices,
                                                             bt.indptr, at)
        else:
            data, indices, indptr = K.dot_csc_ndarray_sparse(bt.shape, tuple(at.shape), bt.data,
                                                             bt.indices, bt.indptr, at)
        # the product was formed transposed, so its compressed axis flips back
        out = GCXS((data, indices, indptr), shape=out_shape, compressed_axes=b.compressed_axes, prune=True)
        return out.tocoo() if rk == "coo" else out

    if isinstance(a, COO) and isinstance(b, COO):
        coords, data = K.dot_coo_coo(out_shape, a.coords, b.coords, a.data, b.data, a.shape[1])
        out = COO(coords, data, shape=out_shape, has_duplicates=False, sorted=True, prune=True)
        if rk == "ndarray":
            return io.out(out.todense_device())
        if rk == "gcxs":
            return out.asformat("gcxs")
        return out

    if isinstance(a, COO) and _is_dense(b):
        bt = dev.to_device(b, a.device)
        if rk in (None, "ndarray"):
            return io.out(_gcxs_times_dense(a, bt, out_shape))
        keys, data = K.dot_coo_ndarray_sparse(a.coords, a.data, bt, out_shape, as_keys=True)
        out = COO._from_sorted_keys(keys, data, out_shape, np.zeros((), dtype=dev.np_dtype(data.dtype))[()], torch.int64)
        return out.asformat("gcxs") if rk == "gcxs" else out

    if _is_dense(a) and isinstance(b, COO):
        at = dev.to_device(a, b.device)
        # b's transpose compressed by rows - what the product runs on - depends on b alone: kept with b (dropped with its other
        # derived layouts when a stored buffer changes).  It doubles b's storage - like the CSR twin of a column-compressed
        # GCXS and the block streams, which are kept at any size; round 6: up to 2^27 stored elements (2^18 until then - a
        # COO of 3 x 10^7 elements was transposed and sorted again at every `dense @ coo`: 4.98 ms against 2.3 ms for a GCXS)
        st = None
        if b.nnz <= COO_T_CSR_MAX_NNZ and hasattr(b, "__dict__"):
            _validate_derived(b)
            st = b.__dict__.get("_csr_of_t")
            if st is None:
                st = b.__dict__["_csr_of_t"] = K.coo_transposed_csr(b.coords, b.data, int(b.shape[0]), int(b.shape[1]))
        if rk in (None, "ndarray"):
            return io.out(K.dot_ndarray_coo(at, b.coords, b.data, out_shape, exact=_settings.EXACT_MULADD, st=st))
        keys, data = K.dot_ndarray_coo_sparse(at, b.coords, b.data, out_shape, as_keys=True, st=st)
        out = COO._from_sorted_keys(keys, data, out_shape, np.zeros((), dtype=dev.np_dtype(data.dtype))[()], torch.int64)
        return out.asformat("gcxs") if rk == "gcxs" else out

    if _is_dense(a) and _is_dense(b):
        if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
            return np.dot(a, b)
        return torch.matmul(dev.to_device(a, io.device), dev.to_device(b, io.device))

    raise TypeError("Unsupported types.")
