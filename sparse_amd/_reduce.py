"""A8: ufunc reductions with implicit fill values (reference `SparseArray.reduce`,
sparse/numba_backend/_sparse_array.py:372-437; `COO._reduce_calc/_reduce_return`,
_coo/core.py:693-723; `_grouped_reduce`, :1631-1661; Appendix D5).

Host logic in Python, arithmetic on the device: kept axes are moved first by a key
permutation, groups are runs of `key // n_cols`, each run is reduced by
`spamd_segment_reduce`, the fold-in of the implicit fill entries is a few elementwise passes.
"""
import numpy as np
import torch

from . import _ffi
from . import _kernels as K
from ._device import code_of, ptr, require_hip, stream_ptr, torch_dtype
from ._utils import equivalent, normalize_axis, prod

_RED_OPS = {"add": 0, "multiply": 1, "maximum": 2, "minimum": 3, "logical_or": 4, "logical_and": 5, "fmax": 6, "fmin": 7}
_SUPER = {"add": np.multiply, "multiply": np.power}


def segment_reduce(data, heads, offs, count, op, want_counts=False, sequential=False):
    """Reduce every run of `data` delimited by `heads` (int64 flags, n+1 entries) with `op`.  `sequential`: always
    one thread per run, strictly left to right (the reference's `sums[j] += ...` order, bit for bit), also when
    the runs are long enough for the faster wave-per-run tree."""
    dev = require_hip(data)
    n = int(data.numel())
    was_bool = data.dtype == torch.bool
    if was_bool:
        # NumPy's boolean add / multiply are logical or / and: the 0/1 bytes must never be summed as integers
        data = data.view(torch.uint8)
        op = {"add": "logical_or", "maximum": "logical_or", "multiply": "logical_and", "minimum": "logical_and"}.get(op, op)
    out = torch.empty(count, dtype=data.dtype, device=dev)
    counts = torch.empty(count, dtype=torch.int64, device=dev) if want_counts else None
    ws = (torch.empty(count + 1, dtype=torch.int64, device=dev)
          if (count and n // max(count, 1) >= 24 and not sequential) else None)
    code = _ffi.U8 if data.dtype == torch.uint8 else code_of(data.dtype)
    _ffi.call("spamd_segment_reduce", _RED_OPS[op], code, n, ptr(data.contiguous()), ptr(heads), ptr(offs), count,
              ptr(out), ptr(counts), ptr(ws), stream_ptr(dev))
    if was_bool:
        out = out.view(torch.bool)
    return (out, counts) if want_counts else out


def group_reduce(keys, divisor, data, op, key_bound=0, sync=True, ng=None):
    """One pass over SORTED keys: runs of equal `keys // divisor` -> (group ids, reduced values, run lengths).
    C ABI `spamd_group_reduce` (reference `_reduce_calc`, _coo/core.py:1601-1661).  `sync=False`: nothing is read back -
    returns the n-sized output buffers and a device int64[2] whose first word is the number of groups."""
    dev = require_hip(keys, data)
    n = int(keys.numel())
    if data.dtype == torch.bool:
        data = data.view(torch.uint8)
    code = _ffi.U8 if data.dtype == torch.uint8 else code_of(data.dtype)
    gids = torch.empty(n, dtype=torch.int64, device=dev)
    vals = torch.empty(n, dtype=data.dtype, device=dev)
    counts = torch.empty(n, dtype=torch.int64, device=dev)
    if ng is None:
        ng = torch.empty(2, dtype=torch.int64, device=dev)
    ws_bytes = int(_ffi.lib().spamd_group_reduce_ws_bytes(code, n))
    if ws_bytes < 0:
        raise _ffi.HipBackendError(f"spamd_group_reduce_ws_bytes failed: {ws_bytes}")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _ffi.call("spamd_group_reduce", _RED_OPS[op], code, n, ptr(keys.contiguous()), int(divisor), int(key_bound),
              ptr(data.contiguous()),
              ptr(gids), ptr(vals), ptr(counts), ptr(ng), ptr(ws), ws_bytes, stream_ptr(dev))
    if not sync:
        return gids, vals, counts, ng
    count = int(ng[0])
    return gids[:count], vals[:count], counts[:count], count


REDUCE_ALL_DIRECT = True
_RA_WS = {}      # (device, stream) -> the zeroed workspace of spamd_reduce_all (its ticket word is left zero by every call)


def reduce_all(data, op):
    """Every element into ONE group without reading keys (C ABI `spamd_reduce_all`; reference _sparse_array.py:372-437 with
    all axes reduced).  Returns device buffers in `group_reduce(..., sync=False)`'s form."""
    dev = require_hip(data)
    if data.dtype == torch.bool:
        data = data.view(torch.uint8)
    code = _ffi.U8 if data.dtype == torch.uint8 else code_of(data.dtype)
    wkey = (dev, stream_ptr(dev))            # concurrent streams must not share the ticket word
    ws = _RA_WS.get(wkey)
    if ws is None:
        ws = _RA_WS[wkey] = torch.zeros(int(_ffi.lib().spamd_reduce_all_ws_bytes()), dtype=torch.uint8, device=dev)
    buf = torch.empty(5, dtype=torch.int64, device=dev)       # one allocation: id, value, count, [groups, equal-to-fill]
    gids, counts, ng = buf[0:1], buf[2:3], buf[3:5]
    vals = buf[1:2].view(data.dtype)[:1]
    data = data.contiguous()
    _ffi.call("spamd_reduce_all", _RED_OPS[op], code, int(data.numel()), ptr(data), ptr(gids), ptr(vals), ptr(counts), ptr(ng),
              ptr(ws), int(ws.numel()), stream_ptr(dev))
    return gids, vals, counts, ng


LONG_RUN_WINDOW = 1024


def has_long_run(offs, window=None):
    """`offs` = the exclusive scan of head flags (n + 1 entries).  True when some aligned window of `window` consecutive
    elements holds no head - a run of at least `window` elements exists, and every run of 2 x window - 1 or more is seen -,
    from every window-th entry of the scan alone (n / 1024 words): what the callers that sum runs with one thread per run
    (left to right, the reference's order) ask before they do, since such a thread walks a long run alone at ~0.2 us per
    element."""
    from ._umath import binary_arrays

    window = window or LONG_RUN_WINDOW
    n = int(offs.numel()) - 1
    if n < window:
        return False
    s = offs[0:n + 1:window].contiguous()
    if s.numel() < 2:
        return False
    same = binary_arrays("equal", s[1:].clone(), s[:-1].contiguous(), out_bool_as=torch.uint8)
    return bool(int(reduce_all(same, "logical_or")[1]))


def _scalar_dev(value, dtype, dev):
    return torch.tensor([value], dtype=dtype, device=dev)


def _device_reducible(x, name, dtype, kwargs):
    """Whether the grouped-reduce kernels cover this reduction: one of the eight table ufuncs, value and result dtypes
    the kernels are instantiated for, no keyword besides `dtype`."""
    if name not in _RED_OPS or kwargs:
        return False
    if x.data.dtype not in K._CODE_T:
        return False
    return dtype is None or torch_dtype_or_none(dtype) in K._CODE_T


def torch_dtype_or_none(dt):
    try:
        return torch_dtype(dt)
    except (KeyError, TypeError):
        return None


def _reduce_on_host(x, method, axis, keepdims, kwargs, out_gcxs):
    """Any other binary ufunc (bitwise_or, hypot, ...), complex or narrow value dtypes, extra keywords: the reference's
    own algorithm (`COO._reduce_calc` + `_grouped_reduce`, _coo/core.py:693-723,1631-1661, and the fill fold-in of
    `SparseArray.reduce`, _sparse_array.py:398-423) with `ufunc.reduceat` evaluated by NumPy on the host over the grouped
    runs.  The STRUCTURE stays on the device: kept axes first by a key permutation, one stable key sort; only the sorted
    keys and values cross to the host, the result container is built on the device again."""
    from ._coo import COO

    kept = tuple(ax for ax in range(x.ndim) if ax not in set(axis))
    n_groups = prod(x.shape[d] for d in kept)
    n_cols = prod(x.shape[d] for d in axis)
    keys, data = x.linear_loc(), x.data
    order = kept + tuple(axis)
    if order != tuple(range(x.ndim)) and x.nnz:
        keys, perm = K.sort_keys(K.permute_keys(keys, x.shape, order), max(x.size - 1, 1))
        data = K.gather(data, perm)
    groups = keys.cpu().numpy() // max(n_cols, 1)
    vals = data.cpu().numpy()
    heads = np.flatnonzero(np.concatenate(([True], groups[1:] != groups[:-1]))) if groups.size else np.empty(0, np.intp)
    counts = np.diff(np.concatenate((heads, [groups.size]))) if groups.size else np.empty(0, np.intp)
    fv = x.fill_value
    super_ufunc = _SUPER.get(getattr(method, "__name__", None))
    with np.errstate(all="ignore"):
        red = method.reduceat(vals, heads, **kwargs)
        result_fill = fv
        if super_ufunc is None:
            missing = counts != n_cols
            red[missing] = method(red[missing], fv, **kwargs)
        else:
            n_fill = n_cols - counts
            contribution = super_ufunc(fv, n_fill)
            if method.identity is not None:   # no implicit entries in the group: the identity, not super(fill, 0)
                contribution = np.where(n_fill == 0, method.identity, contribution)
            red = method(red, contribution).astype(red.dtype)
            result_fill = super_ufunc(fv, n_cols)
    out = COO(groups[heads][None, :].astype(np.int64), red, shape=(n_groups,), has_duplicates=False, sorted=True,
              prune=True, fill_value=result_fill, device=x.device)
    out = out.reshape(tuple(x.shape[d] for d in kept))
    if keepdims:
        shape = list(x.shape)
        for ax in axis:
            shape[ax] = 1
        out = out.reshape(shape)
    if out.ndim == 0:
        return COO.from_numpy(out.todense_device())
    return out.asformat("gcxs") if out_gcxs else out


_SCALAR_PLANS = {}      # (ufunc, value dtype, fill value, dtype keyword) -> the reduction's scalar results (see reduce_impl)


def reduce_impl(x, method, axis=(0,), keepdims=False, _no_merge=False, **kwargs):
    from ._coo import COO
    from ._gcxs import GCXS

    name = getattr(method, "__name__", str(method))
    out_gcxs = isinstance(x, GCXS)
    if out_gcxs:
        # a 2-D GCXS is reduced through the keys of its own compressed layout (kept with it: no conversion to COO per call);
        # compressed_axes = (1,) stores the transpose, so the axes swap
        stand_in = None
        if x.ndim == 2 and x.size and axis is not None and x.compressed_axes in ((0,), (1,)):
            from ._umath import _gcxs_keys2d

            k2 = _gcxs_keys2d(x)
            idt = x.indices.dtype    # the result keeps the operand's index width, as the tocoo() route does
            ax = normalize_axis(axis, 2)
            ax = ax if isinstance(ax, tuple) else (ax,)
            if k2 is not None and x.compressed_axes == (0,):
                stand_in, axis = COO._from_sorted_keys(k2, x.data, x.shape, x.fill_value, idt), ax
            elif k2 is not None and len(ax) == 1 and not keepdims:   # (more axes / keepdims: the result's axes would need swapping back)
                stand_in = COO._from_sorted_keys(k2, x.data, (x.shape[1], x.shape[0]), x.fill_value, idt)
                axis = (1 - ax[0],)
        if stand_in is None and REDUCE_ALL_DIRECT and x.size and (axis is None or set(
                a % x.ndim for a in (axis if isinstance(axis, tuple) else (axis,)) if isinstance(a, int) and -x.ndim <= a < x.ndim
                ) == set(range(x.ndim))) and _device_reducible(x, name, kwargs.get("dtype"), {k: v for k, v in kwargs.items() if k not in ("dtype", "out")}):
            # every axis reduced: only the stored values matter (no coordinates, no keys: `spamd_reduce_all` below)
            stand_in, axis = COO._from_sorted_keys(None, x.data, x.shape, x.fill_value, x.indices.dtype), None
        x = stand_in if stand_in is not None else x.tocoo()
    kwargs.pop("out", None)
    axis = normalize_axis(axis, x.ndim)
    fv = x.fill_value
    super_ufunc = _SUPER.get(name)
    # the scalar arithmetic of a reduction (is op(fill, fill) the fill, the result's dtype, its fill value and their bit
    # patterns) depends on (ufunc, dtypes, fill value, reduced length) only: ~20 us of NumPy scalar work per call, kept
    plan_key = None
    if isinstance(method, np.ufunc) and set(kwargs) <= {"dtype"}:
        fvk = np.asarray(fv)
        plan_key = (name, x.dtype.str, fvk.dtype.str, fvk.tobytes(), str(kwargs.get("dtype")))
    plan = _SCALAR_PLANS.get(plan_key) if plan_key is not None else None
    if plan is None:
        zero_reduce_result = method.reduce([fv, fv], **kwargs)
        dense_result = not equivalent(zero_reduce_result, fv) and super_ufunc is None
        if plan_key is not None:
            plan = {"dense": dense_result}
            if len(_SCALAR_PLANS) > 512:
                _SCALAR_PLANS.clear()
            _SCALAR_PLANS[plan_key] = plan
    else:
        dense_result = plan["dense"]
    if dense_result:
        raise ValueError(f"Performing this reduction operation would produce a dense result: {method!s}")
    dtype = kwargs.pop("dtype", None)
    if not _device_reducible(x, name, dtype, kwargs):
        if dtype is not None:
            kwargs["dtype"] = dtype
        if not isinstance(axis, tuple):
            axis = (axis,)
        if axis == (None,):
            axis = tuple(range(x.ndim))
        return _reduce_on_host(x, method, axis, keepdims, kwargs, out_gcxs)
    if not isinstance(axis, tuple):
        axis = (axis,)
    if axis == (None,):
        axis = tuple(range(x.ndim))
    kept = tuple(ax for ax in range(x.ndim) if ax not in set(axis))
    n_groups = prod(x.shape[d] for d in kept)
    n_cols = prod(x.shape[d] for d in axis)
    dev = x.device

    data = x.data
    if name in ("logical_or", "logical_and"):
        data = K.convert(data, torch.bool)
        res_np_dtype = np.dtype(bool)
    else:
        res_np_dtype = plan.get("res") if plan is not None else None
        if res_np_dtype is None:
            res_np_dtype = np.dtype(dtype) if dtype is not None else (
                method.reduce(np.zeros(1, dtype=x.dtype)).dtype if x.dtype.kind != "b" else np.dtype(bool))
            if x.dtype.kind == "b" and name in ("add", "multiply") and dtype is None:
                res_np_dtype = np.add.reduce(np.zeros(1, dtype=bool)).dtype
            if plan is not None:
                plan["res"] = res_np_dtype
        data = K.convert(data, torch_dtype(res_np_dtype))
    if data.dtype == torch.bool:
        data = data.view(torch.uint8)

    # move kept axes first (key permutation + stable sort), group id = key // n_cols
    direct_all = not kept and REDUCE_ALL_DIRECT
    keys = x.linear_loc() if not direct_all else None
    order = kept + tuple(axis)
    ng = None        # [groups, results equal to the fill value, "the slab merge gave up"]: one read-back below
    if order != tuple(range(x.ndim)) and x.nnz and not direct_all:       # (one group: in whatever order the axes are named)
        merged = None
        if not _no_merge and tuple(axis) == tuple(range(len(axis))) and n_groups > 0:
            # the reduced axes lead: the elements are n_cols sorted runs (one per index of those axes) - merged, not sorted
            ng = torch.zeros(3, dtype=torch.int64, device=dev)
            merged = K.keys_lead_last(keys, data, n_cols, n_groups, ng[2:])
        if merged is not None:
            keys, data = merged
        else:
            ng = None
            keys = K.permute_keys(keys, x.shape, order)
            if data.element_size() in (4, 8):  # the values ride along as the sort payload (no permutation + gather)
                keys, data = K.sort_key_value(keys, data, max(x.size - 1, 1))
            else:
                keys, perm = K.sort_keys(keys, max(x.size - 1, 1))
                data = K.gather(data, perm)
    sc = plan.get(("sc", n_cols)) if plan is not None else None
    if sc is None:
        result_fill = np.asarray(fv).astype(res_np_dtype)[()] if name not in ("logical_or", "logical_and") else np.bool_(fv)
        if super_ufunc is not None:
            with np.errstate(all="ignore"):
                final_fill = np.asarray(super_ufunc(fv, n_cols)).astype(res_np_dtype)[()]
        else:
            final_fill = result_fill
        fvn = np.asarray(result_fill if super_ufunc is None else fv)
        with np.errstate(all="ignore"):
            fill_f = float(fvn.astype(np.float64)) if fvn.dtype.kind != "b" else float(bool(fvn))
            fill_i = int(fvn.astype(res_np_dtype if res_np_dtype.kind in "iu" else np.int64)) if np.isfinite(fill_f) else 0
        eq_np = np.dtype("uint8") if res_np_dtype == np.dtype(bool) else res_np_dtype
        eq_bits = int(np.asarray(final_fill).astype(eq_np).reshape(1).view(f"u{eq_np.itemsize}")[0])
        sc = (result_fill, final_fill, fill_f, fill_i, eq_bits)
        if plan is not None:
            plan[("sc", n_cols)] = sc
    result_fill, final_fill, fill_f, fill_i, eq_bits = sc
    if x.nnz:
        # ONE host read for the whole reduction: the grouped reduce leaves the number of groups on the device, the fold-in
        # of the implicit fill entries (reference :405-422) reads it from there and counts the results that equal the
        # result's fill value, and both numbers come back in a single copy (each `.item()` is a stream synchronisation,
        # which at config-1 sizes costs as much as the kernels)
        n = 1 if direct_all else int(keys.numel())      # the most groups there can be
        vcode = _ffi.U8 if data.dtype == torch.uint8 else code_of(data.dtype)
        if direct_all:   # one group: the keys say nothing
            gids, vals, counts, ng = reduce_all(data, name)
        else:
            gids, vals, counts, ng = group_reduce(keys, max(n_cols, 1), data, name, key_bound=max(int(x.size), 1), sync=False, ng=ng)
        _ffi.call("spamd_reduce_fill_count", _RED_OPS[name], vcode, n, ptr(ng), ptr(vals), ptr(counts), int(n_cols),
                  fill_f, fill_i, eq_bits, ptr(ng) + 8, stream_ptr(dev))
        head = K.read_words(ng)      # (through pinned host memory: no blocking copy)
        count, n_eq = head[0], head[1]
        if len(head) > 2 and head[2]:
            # a cell range (or one output cell) held more elements than the merge kernel's arrays: the general order by sorting
            if dtype is not None:
                kwargs["dtype"] = dtype
            r = reduce_impl(x, method, axis=axis, keepdims=keepdims, _no_merge=True, **kwargs)
            return r.asformat("gcxs") if out_gcxs and not isinstance(r, GCXS) and getattr(r, "ndim", 0) else r
        gids, vals = gids[:count], vals[:count]
        if not kept and not keepdims:
            # everything reduced: a 0-d result, whose value becomes the fill value of an array without stored elements
            # (reference :432-435) - one more 8-byte read instead of a dense 0-d tensor, its scan and its compaction
            value = final_fill
            if count == 1 and not n_eq:
                hv = vals[:1].cpu().numpy()
                value = (hv.view(np.bool_) if res_np_dtype == np.dtype(bool) else hv)[0]
            return COO._from_sorted_keys(torch.empty(0, dtype=torch.int64, device=dev), vals[:0] if vals.dtype != torch.uint8 or
                                         res_np_dtype != np.dtype(bool) else vals[:0].view(torch.bool), (),
                                         np.asarray(value).astype(res_np_dtype)[()], torch.int64)
        if n_eq:    # results equal to the fill value are not stored (rare: a sum that cancels exactly, a max of zeros)
            flags = K.flag_ne_bits(vals, final_fill if vals.dtype != torch.uint8 else np.uint8(bool(final_fill)))
            offs = K.exclusive_scan(flags)
            gids, vals = K.compact(gids, flags, offs, count - n_eq), K.compact(vals, flags, offs, count - n_eq)
    else:
        vals = data[:0]
        gids = torch.empty(0, dtype=torch.int64, device=dev)
    if vals.dtype == torch.uint8 and res_np_dtype == np.dtype(bool):
        vals = vals.view(torch.bool)
    # the group ids ARE the sorted linear keys of the 1-D result: no coordinate matrix, no re-linearisation on the reshape
    out = COO._from_sorted_keys(gids, vals, (n_groups,), final_fill, x._index_dtype)
    out = out.reshape(tuple(x.shape[d] for d in kept))
    if keepdims:
        shape = list(x.shape)
        for ax in axis:
            shape[ax] = 1
        out = out.reshape(shape)
    if out.ndim == 0:
        # 0-d result: the value becomes the fill value of an nnz=0 array (reference :432-435)
        return COO.from_numpy(out.todense_device())
    if out_gcxs:
        return out.asformat("gcxs")
    return out


def var_impl(x, axis=None, dtype=None, ddof=0, keepdims=False):
    """Variance over `axis` (reference `SparseArray.var`, _sparse_array.py:725-814).

    The reference composes it from `sum(keepdims)` / broadcast subtract / square / `sum`, which
    materialises a nearly dense intermediate.  Here the same two-pass formula is evaluated per
    group on the device: var_g = (sum_stored (x - m_g)^2 + n_fill_g * m_g^2) / (n - ddof) with
    m_g = sum_g / n — equal to the reference up to summation order (fp tolerance)."""
    import warnings

    from ._coo import COO
    from ._gcxs import GCXS
    from ._umath import binary_arrays

    out_gcxs = isinstance(x, GCXS)
    if out_gcxs:
        every = axis is None or (isinstance(axis, tuple) and all(isinstance(a, int) and -x.ndim <= a < x.ndim for a in axis)
                                 and set(a % x.ndim for a in axis) == set(range(x.ndim)))
        if every and REDUCE_ALL_DIRECT and x.size and equivalent(x.fill_value, 0, loose=True):
            # every axis reduced: the stored values alone (no coordinates: `spamd_reduce_all` below)
            x = COO._from_sorted_keys(None, x.data, x.shape, x.fill_value, x.indices.dtype)
        else:
            x = x.tocoo()
    if not equivalent(x.fill_value, 0, loose=True):
        # the variance is shift-invariant: var(x) = var(x - fill), and x - fill has a zero background (entries that
        # become exactly 0 drop out of the stored set, which is what they then are)
        x = x - np.asarray(x.fill_value)[()]
    axis = normalize_axis(axis, x.ndim)
    if axis is None:
        axis = tuple(range(x.ndim))
    if not isinstance(axis, tuple):
        axis = (axis,)
    kept = tuple(ax for ax in range(x.ndim) if ax not in set(axis))
    rcount = prod(x.shape[a] for a in axis)
    if ddof >= rcount:
        warnings.warn("Degrees of freedom <= 0 for slice", RuntimeWarning, stacklevel=1)
    if dtype is None:
        dtype = np.dtype("f8") if x.dtype.kind in "iub" else x.dtype
    work = torch_dtype(dtype)
    dev = x.device
    n_groups, n_cols = prod(x.shape[d] for d in kept), rcount
    direct_all = not kept and REDUCE_ALL_DIRECT
    keys, data = (x.linear_loc() if not direct_all else None), K.convert(x.data, work)
    order = kept + tuple(axis)
    if order != tuple(range(x.ndim)) and x.nnz and not direct_all:
        keys = K.permute_keys(keys, x.shape, order)
        if data.element_size() in (4, 8):  # the values ride along as the sort payload (no permutation + gather)
            keys, data = K.sort_key_value(keys, data, max(x.size - 1, 1))
        else:
            keys, perm = K.sort_keys(keys, max(x.size - 1, 1))
            data = K.gather(data, perm)
    denom = max(rcount - ddof, 0)
    if x.nnz:
        # the sums of a group by the grouped reduce (runs of any length: `segment_reduce` gives a long run to ONE wave -
        # 109 ms for the single run of 10^7 elements that axis=None is), or, with every axis reduced, by spamd_reduce_all
        rc = _scalar_dev(rcount, work, dev)
        if direct_all:
            gids, sums, counts, _ = reduce_all(data, "add")
            mean = binary_arrays("divide", sums, rc, b_scalar=True)
            d = binary_arrays("subtract", data, mean, b_scalar=True)
            s1 = reduce_all(binary_arrays("multiply", d, d), "add")[1]
        else:
            kb = max(int(x.size), 1)
            gids, sums, counts, count = group_reduce(keys, max(n_cols, 1), data, "add", key_bound=kb)
            mean = binary_arrays("divide", sums, rc, b_scalar=True)
            gk = binary_arrays("floor_divide_i64", keys, _scalar_dev(max(n_cols, 1), torch.int64, dev), b_scalar=True)
            offs = K.exclusive_scan(K.flag_heads(gk))
            gi = binary_arrays("subtract", offs[1:].contiguous(), _scalar_dev(1, torch.int64, dev), b_scalar=True)  # group index of each element
            d = binary_arrays("subtract", data, K.gather(mean, gi))
            s1 = group_reduce(keys, max(n_cols, 1), binary_arrays("multiply", d, d), "add", key_bound=kb)[1]
        n_fill = K.convert(binary_arrays("subtract", _scalar_dev(n_cols, torch.int64, dev), counts, a_scalar=True), work)
        s = binary_arrays("add", s1, binary_arrays("multiply", n_fill, binary_arrays("multiply", mean, mean)))
        with np.errstate(all="ignore"):
            vals = binary_arrays("divide", s, _scalar_dev(float(denom), work, dev), b_scalar=True)
    else:
        vals = data[:0]
        gids = torch.empty(0, dtype=torch.int64, device=dev)
    out = COO(gids[None, :], vals, shape=(n_groups,), has_duplicates=False, sorted=True, prune=True,
              fill_value=np.dtype(dtype).type(0))
    out = out.reshape(tuple(x.shape[d] for d in kept))
    if keepdims:
        shape = list(x.shape)
        for ax in axis:
            shape[ax] = 1
        out = out.reshape(shape)
    if out.ndim == 0:
        return COO.from_numpy(out.todense_device())
    return out.asformat("gcxs") if out_gcxs else out
