"""The rest of the reference's array namespace around the hot path (`sparse/numba_backend/__init__.py:85-177`): the
coordinate-shuffling functions of `_coo/common.py` and the array-API utilities of `_common.py`.

Nothing here has arithmetic of its own beyond integer coordinate work.  Each function is restated on top of the
library's device primitives - the elementwise kernels over coordinate rows (`_umath.binary_arrays`), flag / scan /
compaction (`_kernels`), the COO constructor's sort, `concatenate`, `broadcast_to`, basic indexing - so operands stay
in HBM; the only host traffic is what the reference itself returns as host values (shapes, dtypes).  Round 4: the
functions that were compositions of other public functions (`repeat`, `tile`, `unstack`, `diff`, `interp`, `take`,
`unique_*`) and the host-side `DOK` container were REMOVED - they are outside SURVEY.md section 8 and carried no device
work of their own."""
import builtins
from collections.abc import Iterable

import numpy as np
import torch

from . import _ffi
from . import _kernels as K
from ._device import ptr, stream_ptr
from ._sparse_array import SparseArray
from ._utils import can_store, check_zero_fill_value, equivalent, normalize_axis, zero_of_dtype


def _is_sparse(x):
    from ._coo import _is_scipy_sparse

    return isinstance(x, SparseArray) or _is_scipy_sparse(x)


def _sc(value, like):
    return torch.tensor([int(value)], dtype=like.dtype, device=like.device)


def _row_op(name, row, value, scalar_first=False):
    """coordinate row (op) integer, on the device"""
    from ._umath import binary_arrays

    row = row.contiguous()
    if scalar_first:
        return binary_arrays(name, _sc(value, row), row, a_scalar=True)
    return binary_arrays(name, row, _sc(value, row), b_scalar=True)


def _iota(n, device):
    t = torch.empty(n, dtype=torch.int64, device=device)
    if n:
        _ffi.call("spamd_iota", n, ptr(t), stream_ptr(device))
    return t


def _keep(mask_u8, coords, data):
    """compact the stored elements whose mask byte is set"""
    flags = K.flag_ne_bits(mask_u8, 0)
    offs = K.exclusive_scan(flags)
    count = int(offs[-1])
    return K.compact(coords, flags, offs, count), K.compact(data, flags, offs, count)


# ---- conversions and dtype helpers --------------------------------------------------------------------------------

def asCOO(x, name="asCOO", check=True):
    """`_coo/common.py:21-53`: any sparse array as COO; a dense input is refused unless `check=False`."""
    from ._coo import COO

    if check and not _is_sparse(x):
        raise ValueError(f"Performing this operation would produce a dense result: {name}")
    if not isinstance(x, COO):
        x = COO(x)
    return x


def _validate_coo_input(x):
    """`_coo/common.py:1386-1397`"""
    from ._coo import COO, _is_scipy_sparse

    if _is_scipy_sparse(x):
        return COO.from_scipy_sparse(x)
    if not isinstance(x, SparseArray):
        raise ValueError(f"Input must be an instance of SparseArray, but it's {type(x)}.")
    return x if isinstance(x, COO) else x.asformat("coo")


def asnumpy(a, dtype=None, order=None):
    """`_common.py:1928-1948`: the dense host array"""
    if isinstance(a, SparseArray):
        a = a.todense()
    if isinstance(a, torch.Tensor):
        a = a.cpu().numpy()
    return np.asarray(a, dtype=dtype, order=order)


def can_cast(from_, to, /, *, casting="safe"):
    """`_common.py:1863-1892`"""
    return np.can_cast(np.dtype(getattr(from_, "dtype", from_)), to, casting=casting)


def result_type(*arrays_and_dtypes):
    """`_coo/common.py:991-1008`: sparse arrays count by dtype (0-d ones by value)."""
    def arg(x):
        if not isinstance(x, SparseArray):
            return x
        return x.dtype if x.ndim > 0 else x.todense()

    return np.result_type(*(arg(x) for x in arrays_and_dtypes))


def broadcast_shapes(*shapes):
    return np.broadcast_shapes(*shapes)


def broadcast_arrays(*arrays):
    """`_common.py:2835-2869`"""
    from ._coo import COO

    shape = np.broadcast_shapes(*[a.shape for a in arrays])
    out = []
    for a in arrays:
        if isinstance(a, (np.generic, np.ndarray)):
            a = COO.from_numpy(a)
        out.append(a.broadcast_to(shape))
    return tuple(out)


def _numpy_first(name):
    """The reference's `_support_numpy` (`_common.py:2139-2159`): a dense first argument goes to NumPy, with a warning."""
    import functools
    import warnings

    def deco(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            if isinstance(args[0], (np.ndarray, np.number)):
                warnings.warn(f"Sparse {name} received dense NumPy array instead of sparse array. Dispatching to NumPy function.",
                              RuntimeWarning, stacklevel=2)
                return getattr(np, name)(*args, **kwargs)
            return func(*args, **kwargs)

        return wrapper

    return deco


@_numpy_first("round")
def round(x, /, decimals=0, out=None):  # noqa: A001 - the reference's name
    return x.round(decimals=decimals, out=out)


@_numpy_first("isinf")
def isinf(x, /):
    return x.isinf()


@_numpy_first("isnan")
def isnan(x, /):
    return x.isnan()


def abs(x, /):  # noqa: A001
    return x.__abs__()


def equal(x1, x2, /):
    return x1 == x2


def real(x, /):
    return x.real


def imag(x, /):
    return x.imag


def clip(a, min=None, max=None, out=None):  # noqa: A002
    """`_coo/common.py:1028-1071`"""
    return asCOO(a, name="clip").clip(min, max)


def _same_sign_inf(x, positive):
    """isposinf / isneginf as a device comparison with the infinity itself (NaN compares false, integers never match);
    NumPy refuses complex input here, so does this."""
    from ._umath import elemwise

    if not _is_sparse(x):
        return (np.isposinf if positive else np.isneginf)(x)
    if np.issubdtype(x.dtype, np.complexfloating):
        raise TypeError("This operation is not supported for complex values because it would be ambiguous.")
    if np.issubdtype(x.dtype, np.floating):
        return elemwise(np.equal, x, x.dtype.type(np.inf if positive else -np.inf))
    return elemwise(np.not_equal, x, x)     # integers and booleans: nowhere


def isposinf(x, out=None):
    """`_coo/common.py:937-961`"""
    if out is not None:
        raise NotImplementedError("`out=` is not supported here")
    return _same_sign_inf(x, True)


def isneginf(x, out=None):
    """`_coo/common.py:964-988`"""
    if out is not None:
        raise NotImplementedError("`out=` is not supported here")
    return _same_sign_inf(x, False)


# ---- coordinate shuffles --------------------------------------------------------------------------------------------

def flip(x, /, *, axis=None):
    """`_coo/common.py:1136-1178`: coordinate c of a flipped axis becomes n - 1 - c."""
    from ._coo import COO

    x = _validate_coo_input(x)
    if axis is None:
        axis = range(x.ndim)
    if not isinstance(axis, Iterable):
        axis = (axis,)
    rows = [x.coords[d] for d in range(x.ndim)]
    for ax in axis:
        if x.nnz:
            rows[ax] = _row_op("subtract", rows[ax], x.shape[ax] - 1, scalar_first=True)
    coords = torch.stack(rows) if rows else x.coords
    return COO(coords, x.data, shape=x.shape, fill_value=x.fill_value)


def roll(a, shift, axis=None):
    """`_coo/common.py:735-812`: coordinates move by `shift` modulo the axis length."""
    from ._coo import COO, as_coo

    a = as_coo(a)
    if axis is None:
        return roll(a.reshape((-1,)), shift, 0).reshape(a.shape)
    axis = normalize_axis(axis, a.ndim)
    if not isinstance(axis, tuple):
        axis = (axis,)
    if not isinstance(shift, Iterable):
        shift = (shift,)
    elif np.ndim(shift) > 1:
        raise ValueError("'shift' and 'axis' must be integers or 1D sequences.")
    if len(shift) == 1:
        shift = np.full(len(axis), shift)
    if len(axis) != len(shift):
        raise ValueError("If 'shift' is a 1D sequence, 'axis' must have equal length.")
    idx_np = np.dtype(str(a.coords.dtype).replace("torch.", ""))
    if not can_store(idx_np, builtins.max(a.shape + shift)):       # (the reference's own expression, broadcasting quirk included)
        raise ValueError(f"cannot roll with coords.dtype {idx_np} and shift {shift}. Try casting coords to a larger dtype.")
    rows = [a.coords[d] for d in range(a.ndim)]
    for sh, ax in zip(shift, axis, strict=True):
        n = a.shape[ax]
        if not a.nnz or n == 0:
            continue
        moved = _row_op("add", rows[ax], int(sh) % n)           # in [0, 2 n)
        over = K.convert(_row_op("greater_equal", moved, n).view(torch.uint8), moved.dtype)
        rows[ax] = _binary("subtract", moved, _row_op("multiply", over, n))
    coords = torch.stack(rows) if rows else a.coords
    return COO(coords, data=a.data.clone(), shape=a.shape, has_duplicates=False, fill_value=a.fill_value)


def _binary(name, a, b):
    from ._umath import binary_arrays

    return binary_arrays(name, a.contiguous(), b.contiguous())


def pad(array, pad_width, mode="constant", **kwargs):
    """`_common.py:2002-2061`: constant padding with the fill value = shifted coordinates in a larger shape."""
    from ._coo import COO

    if not isinstance(array, SparseArray):
        raise NotImplementedError("Input array is not compatible.")
    if mode.lower() != "constant":
        raise NotImplementedError(f"Mode '{mode}' is not yet supported.")
    if not equivalent(kwargs.pop("constant_values", zero_of_dtype(array.dtype)), array.fill_value):
        raise ValueError("constant_values can only be equal to fill value.")
    if kwargs:
        raise NotImplementedError("Additional Unknown arguments present.")
    array = array.asformat("coo")
    pad_width = np.broadcast_to(pad_width, (len(array.shape), 2))
    rows = [array.coords[d] for d in range(array.ndim)]
    if array.nnz:
        rows = [_row_op("add", r, int(pad_width[d, 0])) if int(pad_width[d, 0]) else r for d, r in enumerate(rows)]
    new_shape = tuple(array.shape[i] + int(pad_width[i, 0]) + int(pad_width[i, 1]) for i in range(len(array.shape)))
    coords = torch.stack(rows) if rows else array.coords
    return COO(coords, array.data, new_shape, fill_value=array.fill_value)


def _triangle(x, k, upper):
    from ._coo import COO

    check_zero_fill_value(x)
    if not x.ndim >= 2:
        raise NotImplementedError(f"sparse.{'triu' if upper else 'tril'} is not implemented for scalars or 1-D arrays.")
    x = _validate_coo_input(x) if not hasattr(x, "coords") else x
    if not x.nnz:
        return COO(x.coords, x.data, shape=x.shape, has_duplicates=False, sorted=True)
    lhs = _row_op("add", x.coords[-2], k)
    mask = _binary("less_equal" if upper else "greater_equal", lhs, x.coords[-1]).view(torch.uint8)
    coords, data = _keep(mask, x.coords, x.data)
    return COO(coords, data, shape=x.shape, has_duplicates=False, sorted=True)


def triu(x, k=0):
    """`_coo/common.py:252-290`: elements with row + k <= column of the last two axes."""
    return _triangle(x, k, True)


def tril(x, k=0):
    """`_coo/common.py:293-331`: elements with row + k >= column of the last two axes."""
    return _triangle(x, k, False)


def diagonal(a, offset=0, axis1=0, axis2=1):
    """`_coo/common.py:815-878` (`_diagonal_idx` :1012-1025): stored elements with coords[axis1] + offset == coords[axis2]; the result's last axis
    carries coords[axis1] (the reference's choice, also for negative offsets)."""
    from ._coo import COO

    if a.shape[axis1] != a.shape[axis2]:
        raise ValueError("a.shape[axis1] != a.shape[axis2]")
    a = _validate_coo_input(a)
    diag_axes = [ax for ax in range(len(a.shape)) if ax != axis1 and ax != axis2] + [axis1]
    diag_shape = [a.shape[ax] for ax in diag_axes]
    diag_shape[-1] -= builtins.abs(offset)
    if a.nnz:
        mask = _binary("equal", _row_op("add", a.coords[axis1], offset), a.coords[axis2]).view(torch.uint8)
        coords, data = _keep(mask, a.coords, a.data)
    else:
        coords, data = a.coords, a.data
    return COO(torch.stack([coords[ax] for ax in diag_axes]), data, diag_shape)


def diagonalize(a, axis=0):
    """`_coo/common.py:881-934`: a new last axis that repeats the coordinate of `axis`."""
    from ._coo import COO, as_coo

    a = as_coo(a)
    coords = torch.cat([a.coords, a.coords[axis][None, :]], dim=0)
    return COO(coords, a.data, a.shape + (a.shape[axis],))


def kron(a, b):
    """`_coo/common.py:67-129`: every pair (stored element of a, stored element of b); coordinate = a's times b's extent
    plus b's, value = the product."""
    from ._coo import COO

    check_zero_fill_value(a, b)
    a_sparse, b_sparse = _is_sparse(a), _is_sparse(b)
    a_ndim, b_ndim = np.ndim(a), np.ndim(b)
    if not (a_sparse or b_sparse):
        raise ValueError("Performing this operation would produce a dense result: kron")
    if a_ndim == 0 or b_ndim == 0:
        return a * b
    a, b = asCOO(a, check=False), asCOO(b, check=False)
    nd = builtins.max(a.ndim, b.ndim)
    a = a.reshape((1,) * (nd - a.ndim) + a.shape)
    b = b.reshape((1,) * (nd - b.ndim) + b.shape)
    na, nb = a.nnz, b.nnz
    o_shape = tuple(i * j for i, j in zip(a.shape, b.shape, strict=True))
    from ._device import torch_dtype

    tdt = torch_dtype(np.result_type(a.dtype, b.dtype))          # NumPy's `a.data[i] * b.data[j]` dtype for two arrays
    if na * nb == 0:
        return COO(torch.zeros((nd, 0), dtype=a.coords.dtype, device=a.device), torch.zeros(0, dtype=tdt, device=a.device),
                   shape=o_shape, has_duplicates=False)
    pair = _iota(na * nb, a.device)
    ia = _row_op("floor_divide_i64", pair, nb)                      # pair // nb
    ib = _binary("subtract", pair, _row_op("multiply", ia, nb))     # pair % nb
    wide = torch.int64
    rows = []
    for d in range(nd):
        ca = K.gather(K.convert(a.coords[d].contiguous(), wide), ia)
        cb = K.gather(K.convert(b.coords[d].contiguous(), wide), ib)
        rows.append(_binary("add", _row_op("multiply", ca, b.shape[d]), cb))
    from ._umath import binary_arrays

    va = K.gather(K.convert(a.data.contiguous(), tdt), ia)
    vb = K.gather(K.convert(b.data.contiguous(), tdt), ib)
    name = "logical_and" if tdt == torch.bool else "multiply"
    data = binary_arrays(name, va, vb)
    return COO(torch.stack(rows), data, shape=o_shape, has_duplicates=False)


def outer(a, b, out=None):
    """`_common.py:1895-1925`"""
    from ._coo import COO

    if isinstance(a, SparseArray):
        a = COO(a)
    if isinstance(b, SparseArray):
        b = COO(b)
    return np.multiply.outer(a.flatten(), b.flatten(), out=out)


# ---- compositions of reshape / broadcast / indexing / concatenate ----------------------------------------------------

def concat(arrays, axis=0, compressed_axes=None):
    from ._batched import concatenate

    return concatenate(arrays, axis=axis, compressed_axes=compressed_axes)


# ---- order statistics: argmax / argmin, sort, unique ------------------------------------------------------------------

def _where(mask, a, b, a_scalar=False, b_scalar=False):
    """where(mask, a, b) over device arrays of one dtype (1-element arrays with *_scalar)"""
    m = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    n = int(m.numel())
    out = torch.empty(n, dtype=a.dtype, device=m.device)
    if n:
        _ffi.call("spamd_ewise_select", out.element_size(), n, ptr(m.contiguous()), ptr(a.contiguous()), int(a_scalar),
                  ptr(b.contiguous()), int(b_scalar), ptr(out), stream_ptr(m.device))
    return out


def _i64(value, device):
    return torch.tensor([int(value)], dtype=torch.int64, device=device)


def _group_ordinals(gid):
    """For sorted group ids: (ordinal of every element's group, number of groups is ordinal[-1] + 1)"""
    heads = K.flag_heads(gid.contiguous())
    offs = K.exclusive_scan(heads)
    return _row_op("subtract", offs[1:].contiguous(), 1)


def _order_keys(data, descending=False):
    """int64 keys whose SIGNED order is NumPy's sort order of the values (NaN last; first when descending)."""
    from ._umath import binary_arrays, unary_array

    dev = data.device
    if data.dtype in (torch.float32, torch.float64):
        v = K.convert(data.contiguous(), torch.float64)                  # exact
        b = v.view(torch.int64)
        # negative floats: larger magnitude = smaller value, so every bit below the sign is flipped
        neg = binary_arrays("less", b, _i64(0, dev), b_scalar=True)
        keys = binary_arrays("bitwise_xor", b, _where(neg, _i64(2 ** 63 - 1, dev), _i64(0, dev), a_scalar=True, b_scalar=True))
        nan = unary_array("isnan", v)
        keys = _where(nan, _i64(2 ** 63 - 1, dev), keys, a_scalar=True)
    elif data.dtype in (torch.int32, torch.int64, torch.bool, torch.uint8):
        keys = K.convert(data.contiguous(), torch.int64)
    else:
        raise TypeError(f"the hip backend orders float32 / float64 / int32 / int64 / bool values, not {data.dtype}")
    if descending:
        keys = binary_arrays("bitwise_xor", keys, _i64(-1, dev), b_scalar=True)
    return keys


ALL_KEY_BITS = 2 ** 64 - 1      # sort on all 64 bits: rocPRIM's codec then orders int64 keys as SIGNED numbers


def _arg_minmax(x, axis, keepdims, max_mode):
    """`_coo/common.py:1499-1568` and its kernel `_compute_minmax_args` (:1455-1496) per output position:
    the coordinate of the first best STORED value if one beats the fill value (or nothing but stored values is there),
    otherwise the first coordinate that holds no stored value."""
    from ._coo import COO
    from ._reduce import group_reduce
    from ._umath import binary_arrays

    x = _validate_coo_input(x)
    if not isinstance(axis, (int, type(None))):
        raise ValueError(f"`axis` must be `int` or `None`, but it's: {type(axis)}.")
    if isinstance(axis, int) and axis >= x.ndim:
        raise ValueError(f"`axis={axis}` is out of bounds for array of dimension {x.ndim}.")
    if x.ndim == 0:
        raise ValueError("Input array must be at least 1-D, but it's 0-D.")
    none_ndim = None
    if axis is None:
        none_ndim = x.ndim
        x = x.reshape(-1)[:, None]
        axis = 0
    if axis == 0 and x.ndim == 1:
        x = x[:, None]
    order = list(range(x.ndim))
    order.insert(0, order.pop(axis))
    new_shape = list(x.shape)
    new_shape.insert(0, new_shape.pop(axis))
    R = int(new_shape[0])
    C = int(np.prod(new_shape[1:], dtype=np.int64))
    # (reduced axis LAST: stored elements grouped by output position, reduced coordinate ascending inside a group)
    xt = x.transpose(tuple(order[1:]) + (order[0],)).reshape((C, R))
    dev = xt.device
    n = xt.nnz
    if n:
        if xt.data.dtype not in (torch.float32, torch.float64, torch.int32, torch.int64):
            raise TypeError(f"argmax / argmin of {xt.dtype} values is not covered by the hip backend")
        keys = xt.linear_loc()
        op = "maximum" if max_mode else "minimum"
        gids, best, counts, ng = group_reduce(keys, R, xt.data, op, key_bound=C * R)
        ordinal = _group_ordinals(binary_arrays("floor_divide_i64", keys, _i64(R, dev), b_scalar=True))
        r = _binary("subtract", keys, _row_op("multiply", K.gather(gids, ordinal), R))
        # first stored element that IS the group's best (NaN counts as best, as for numpy.argmax / argmin)
        best_e = K.gather(best, ordinal)
        from ._umath import unary_array

        is_best = binary_arrays("equal", xt.data.contiguous(), best_e, out_bool_as=torch.uint8)
        if xt.data.dtype in (torch.float32, torch.float64):
            both_nan = binary_arrays("logical_and", unary_array("isnan", xt.data.contiguous()).view(torch.uint8),
                                     unary_array("isnan", best_e).view(torch.uint8), out_bool_as=torch.uint8)
            is_best = binary_arrays("logical_or", is_best, both_nan, out_bool_as=torch.uint8)
        cand = _where(is_best, r, _i64(R, dev), b_scalar=True)
        _, first_best, _, _ = group_reduce(keys, R, cand, "minimum", key_bound=C * R)
        # first coordinate without a stored element: position j inside the group whose coordinate is not j
        starts = K.exclusive_scan(torch.cat([counts, counts[:1]]))[:-1]
        j = _binary("subtract", _iota(n, dev), K.gather(starts.contiguous(), ordinal))
        gap = _where(binary_arrays("not_equal", r, j, out_bool_as=torch.uint8), j, _i64(R, dev), b_scalar=True)
        _, first_gap, _, _ = group_reduce(keys, R, gap, "minimum", key_bound=C * R)
        first_gap = _binary("minimum", first_gap.contiguous(), counts.contiguous())
        # does any stored value beat the fill value?
        fill = torch.tensor([xt.fill_value], dtype=xt.data.dtype, device=dev)
        beats = binary_arrays("greater" if max_mode else "less", xt.data.contiguous(), fill, b_scalar=True, out_bool_as=torch.uint8)
        _, any_beats, _, _ = group_reduce(keys, R, K.convert(beats, torch.int64), "maximum", key_bound=C * R)
        full = binary_arrays("equal", counts.contiguous(), _i64(R, dev), b_scalar=True, out_bool_as=torch.uint8)
        use_stored = binary_arrays("logical_or", binary_arrays("not_equal", any_beats.contiguous(), _i64(0, dev), b_scalar=True,
                                                               out_bool_as=torch.uint8), full, out_bool_as=torch.uint8)
        result = _where(use_stored, first_best.contiguous(), first_gap)
        out = COO(gids[None, :], result, shape=(C,), fill_value=0, prune=True, has_duplicates=False, sorted=True)
    else:
        out = COO(torch.zeros((1, 0), dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), shape=(C,),
                  fill_value=0)
    out = out.reshape((1, *new_shape[1:]))
    back = list(range(out.ndim))
    back.insert(axis, back.pop(0))
    out = out.transpose(back)
    if none_ndim is not None:
        out = out.reshape([1 for _ in range(none_ndim)])
    return out if keepdims else out.squeeze()


def argmax(x, /, *, axis=None, keepdims=False):
    """`_coo/common.py:614-641`"""
    return _arg_minmax(x, axis, keepdims, True)


def argmin(x, /, *, axis=None, keepdims=False):
    """`_coo/common.py:644-671`"""
    return _arg_minmax(x, axis, keepdims, False)


def sort(x, /, *, axis=-1, descending=False, stable=False):
    """`_coo/common.py:1280-1346` and `_sort_coo` (:1401-1451): along `axis` the stored values of every line are sorted and
    take the leading coordinates; from the first value the fill value is smaller than (larger than, descending) onwards
    they move behind the line's fill values."""
    from ._api import moveaxis
    from ._coo import COO
    from ._umath import binary_arrays

    x = _validate_coo_input(x)
    if stable:
        raise ValueError("`stable=True` isn't currently supported.")
    original_ndim = x.ndim
    if x.ndim == 1:
        x = x[None, :]
        axis = -1
    x = moveaxis(x, source=axis, destination=-1)
    x_shape = x.shape
    L = int(x_shape[-1])
    x = x.reshape((-1, L))
    G = int(x.shape[0])
    n, dev = x.nnz, x.device
    if n:
        keys = x.linear_loc()
        g = binary_arrays("floor_divide_i64", keys, _i64(L, dev), b_scalar=True)
        # two stable sorts: by value, then by line
        _, p1 = K.sort_keys(_order_keys(x.data, descending), ALL_KEY_BITS)
        g_sorted, p2 = K.sort_keys(K.gather(g, p1), builtins.max(G - 1, 1))
        perm = K.gather(p1, p2)
        data = K.gather(x.data.contiguous(), perm)
        ordinal = _group_ordinals(g_sorted)
        heads = K.flag_heads(g_sorted.contiguous())
        hoffs = K.exclusive_scan(heads)
        n_groups = int(hoffs[-1])
        starts = K.compact(_iota(n, dev), heads, hoffs, n_groups)                        # first element of every line
        ends = torch.cat([starts[1:], _i64(n, dev)])
        counts = _binary("subtract", ends, starts)
        pos = _binary("subtract", _iota(n, dev), K.gather(starts, ordinal))
        fill = torch.tensor([x.fill_value], dtype=data.dtype, device=dev)
        # the reference's `fill_value < data[pos]` (descending: `>`), then everything from the first hit on
        hit = binary_arrays("greater" if not descending else "less", data, fill, b_scalar=True, out_bool_as=torch.uint8)
        hflags = torch.cat([K.convert(hit, torch.int64), _i64(0, dev)])
        hsum = K.exclusive_scan(hflags)
        upto = _binary("subtract", hsum[1:].contiguous(), K.gather(hsum, K.gather(starts, ordinal)))
        moved = K.convert(binary_arrays("greater", upto, _i64(0, dev), b_scalar=True, out_bool_as=torch.uint8), torch.int64)
        room = _row_op("subtract", counts, L, scalar_first=True)                        # fill values of the line
        new_s = _binary("add", pos, _binary("multiply", moved, K.gather(room, ordinal)))
        coords = torch.stack([g_sorted, new_s])
    else:
        coords, data = x.coords, x.data
    out = COO(coords, data, x.shape, has_duplicates=False, sorted=True, fill_value=x.fill_value)
    out = out.reshape(x_shape[:-1] + (x_shape[-1],))
    out = moveaxis(out, source=-1, destination=axis)
    if original_ndim == out.ndim:
        return out
    out = out.squeeze()
    if out.shape == ():
        return out[None]
    return out


