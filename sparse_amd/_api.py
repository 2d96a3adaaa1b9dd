"""Function surface of the hip backend on the hot path (reference
sparse/numba_backend/__init__.py:179-350, the subset SURVEY.md §8b names): thin wrappers that
delegate to the containers, like the reference's own (`_common.py:2077-2633`)."""
import builtins

import numpy as np
import torch

from . import _device as dev
from . import _ffi
from . import _kernels as K
from ._coo import COO, as_coo
from ._gcxs import GCXS
from ._sparse_array import SparseArray


def asarray(obj, /, *, dtype=None, format="coo", copy=False, device=None):
    """Convert to a sparse array of this backend (reference `asarray`, _common.py:2077-2135)."""
    if format not in {"coo", "gcxs"}:
        raise ValueError(f"{format} format not supported.")
    if isinstance(obj, SparseArray):
        out = obj.asformat(format)
    elif hasattr(obj, "tocoo") and type(obj).__module__.startswith("scipy.sparse"):
        out = (GCXS.from_scipy_sparse(obj, device=device) if format == "gcxs" else COO.from_scipy_sparse(obj, device=device))
    elif np.isscalar(obj) or isinstance(obj, (np.ndarray, torch.Tensor, list, tuple)):
        arr = obj if isinstance(obj, torch.Tensor) else np.asarray(obj)
        out = COO.from_numpy(arr, device=device).asformat(format)
    else:
        raise ValueError(f"{type(obj)} not supported.")
    if dtype is not None and out.dtype != np.dtype(dtype):
        out = out.astype(dtype)
    return out


def sddmm(s, a, b=None, *, bt=None):
    """Sampled dense-dense matmul: `s * (a @ b)` evaluated only at the stored positions of the
    2-D sparse mask `s` (the reference's formulation, examples/sddmm_example.py:51-52, forms the
    whole dense product first).  Pass `b` (K x N) or its transpose `bt` (N x K, K-contiguous:
    avoids a device transpose).  Dense operands may be bfloat16/float32/float64 torch tensors
    (bf16/fp32 accumulate in fp32) or float ndarrays.  Result has the format of `s`, zeros pruned."""
    from ._dot import _validate_derived
    from ._utils import check_zero_fill_value

    check_zero_fill_value(s)
    if s.ndim != 2:
        raise ValueError("sddmm needs a 2-D sparse mask")
    out_gcxs = isinstance(s, GCXS)
    if isinstance(s, COO):
        sc = s
    else:  # the COO view of a GCXS mask is kept on it, and the plans on the view
        _validate_derived(s)
        sc = s.__dict__.get("_coo_view")
        if sc is None:
            sc = s.__dict__["_coo_view"] = s.tocoo()
    at = dev.to_device(a, sc.device)
    if (b is None) == (bt is None):
        raise ValueError("pass exactly one of b / bt")
    btt = dev.to_device(bt, sc.device) if bt is not None else dev.to_device(b, sc.device).t().contiguous()
    if at.shape[0] != s.shape[0] or btt.shape[0] != s.shape[1] or at.shape[1] != btt.shape[1]:
        raise ValueError("shape-mismatch for sum")
    at, btt = K.sddmm_pad_inner(at, btt, sc.nnz)
    _validate_derived(sc)
    # plans depend on the pattern only and are kept on the mask (dropped with its other derived layouts when the
    # coordinates change): the populated 32 x 32 tiles that go to the matrix cores, and - when Bt is larger than an
    # XCD's L2 - the column-panel order of whatever the sampled kernel takes
    width = K.sddmm_panel_width(btt)
    plans = sc.__dict__.setdefault("_sddmm_plan", {})

    def panels_of(subset, tag):
        key = ("panels", tag, width)
        if key not in plans:
            plans[key] = K.sddmm_panels(sc.coords, sc.shape, width, subset=subset)
        return plans[key]

    vals = None
    if at.dtype == torch.bfloat16 and btt.dtype == torch.bfloat16 and sc.nnz >= K.SDDMM_TILE_THRESHOLD and at.shape[1] % 16 == 0:
        key = ("tiles", K.SDDMM_TILE_THRESHOLD)
        if key not in plans:
            plans[key] = K.sddmm_plan(sc.coords, sc.shape)
        plan = plans[key]
        if K.sddmm_tiles_pay(plan, at, btt, width):
            # the left-over samples: in panel order if that pays for so many, else as ONE panel (= the mask's own order;
            # either way their coordinates are gathered once, here, and the kernel writes to their positions)
            nrest = int(plan.rest.numel())
            rest = None
            if nrest and K.sddmm_has_panels(at.dtype, at.shape[1]):
                rw = width if K.sddmm_panels_pay(nrest, at, btt, width) else builtins.max(int(sc.shape[1]), 1)
                key = ("panels", "rest", rw)
                if key not in plans:
                    plans[key] = K.sddmm_panels(sc.coords, sc.shape, rw, subset=plan.rest)
                rest = plans[key]
            vals = K.sddmm_coo_mfma(plan, sc.coords, sc.shape, sc.data, at, btt, force=True, rest_panels=rest)
    if vals is None:
        vals = K.sddmm_coo(sc.coords, sc.data, at, btt,
                           panels=panels_of(None, "all") if K.sddmm_panels_pay(sc.nnz, at, btt, width) else None)
    out = COO(sc.coords, vals, shape=s.shape, has_duplicates=False, sorted=True, prune=True)
    return out.asformat("gcxs", compressed_axes=s.compressed_axes) if out_gcxs else out


def random(shape, density=None, nnz=None, random_state=None, data_rvs=None, format="coo", fill_value=None,
           idx_dtype=None, dtype=None, device=None, **kwargs):
    """Random sparse array generated ON THE DEVICE: uniform-without-replacement positions,
    values U[0, 1) — the distribution of the reference's `sparse.random` (_utils.py:221-346; its
    Vitter sampler is a sequential loop).  The random stream is torch's, not NumPy's: same
    distribution, different numbers."""
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
    size = 1
    for s in shape:
        size *= s
    if density is not None and nnz is not None:
        raise ValueError("Specify either density or nnz, not both")
    if density is None and nnz is None:
        density = 0.01
    if density is not None and not 0 <= density <= 1:
        raise ValueError(f"density {density} is not in the unit interval")
    if nnz is None:
        nnz = int(size * density)
    if not 0 <= nnz <= size:
        raise ValueError(f"Cannot generate {nnz} nonzero elements for an array with {size} total elements.")
    d = torch.device(device) if device is not None else dev.default_device()
    g = torch.Generator(device=d)
    if random_state is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0])   # fresh entropy; torch's global RNG is left alone
    elif isinstance(random_state, np.random.Generator):
        seed = int(random_state.integers(2 ** 31))
    elif isinstance(random_state, np.random.RandomState):
        seed = int(random_state.randint(2 ** 31))
    else:
        seed = int(random_state)
    g.manual_seed(seed % (2 ** 63))
    if nnz > size // 2 and size <= 2 ** 31:
        keys = torch.randperm(size, generator=g, device=d)[:nnz].sort().values
    else:
        keys = torch.empty(0, dtype=torch.int64, device=d)
        while keys.numel() < nnz:
            need = nnz - keys.numel()
            draw = torch.randint(0, builtins.max(size, 1), (int(need * 1.05) + 1024,), generator=g, device=d)
            keys = torch.unique(torch.cat([keys, draw]))
        if keys.numel() > nnz:
            keep = torch.ones(keys.numel(), dtype=torch.bool, device=d)
            keep[torch.randperm(keys.numel(), generator=g, device=d)[: keys.numel() - nnz]] = False
            keys = keys[keep]
    if data_rvs is not None:
        # the reference's contract (_utils.py:313-322): data_rvs(nnz) returns the stored values; evaluated on the host
        vals = np.asarray(data_rvs(nnz))
        if vals.shape != (nnz,):
            raise ValueError("data_rvs must return an array of length nnz")
        data = torch.from_numpy(np.ascontiguousarray(vals if dtype is None else vals.astype(dtype))).to(d)
    else:
        data = torch.rand(nnz, generator=g, device=d, dtype=torch.float64).to(dev.torch_dtype(dtype or np.float64))
    it = torch.int64 if idx_dtype is None or np.dtype(idx_dtype).itemsize > 4 else torch.int32
    coords = K.delinearize(keys.contiguous(), shape, it)
    out = COO(coords, data, shape=shape, has_duplicates=False, sorted=True, fill_value=fill_value)
    out._keys = keys
    return out.asformat(format, **kwargs) if format != "coo" else out


def _reduction(name):
    def f(x, /, *, axis=None, keepdims=False, **kw):
        return getattr(x, name)(axis=axis, keepdims=keepdims, **kw)

    f.__name__ = name
    return f


sum = _reduction("sum")
prod = _reduction("prod")
max = _reduction("max")
min = _reduction("min")
mean = _reduction("mean")
any = _reduction("any")
all = _reduction("all")
var = _reduction("var")
std = _reduction("std")


def astype(x, dtype, /, *, copy=True):
    return x.astype(dtype, copy=copy)


def reshape(x, /, shape, *, copy=None):
    return x.reshape(shape)


def permute_dims(x, /, axes=None):
    return x.transpose(axes)


def matrix_transpose(x, /):
    return x.mT


def vecdot(x1, x2, /, *, axis=-1):
    """sum(x1 * x2, axis) (reference `vecdot`, _common.py)."""
    return (x1 * x2).sum(axis=axis)


# ---- creation (reference _common.py:1561-1857: arrays with no stored element and the value as fill value) ----------
def full(shape, fill_value, dtype=None, format="coo", order="C", *, device=None, **kwargs):
    """Array of `shape` whose every element is `fill_value`: nnz = 0, the value is the fill value (reference
    `full`, _common.py:1628-1681)."""
    if dtype is None:
        dtype = np.array(fill_value).dtype
    if not isinstance(shape, tuple):
        shape = (shape,)
    if order not in {"C", None}:
        raise NotImplementedError("Currently, only 'C' and None are supported.")
    d = torch.device(device) if device is not None else dev.default_device()
    out = COO(torch.zeros((len(shape), 0), dtype=torch.int64, device=d),
              torch.zeros(0, dtype=dev.torch_dtype(np.dtype(dtype)), device=d), shape=shape,
              fill_value=np.asarray(fill_value).astype(dtype)[()], has_duplicates=False, sorted=True)
    return out.asformat(format, **kwargs) if format != "coo" else out


def full_like(a, fill_value, dtype=None, shape=None, format=None, *, device=None, **kwargs):
    if format is None and not isinstance(a, np.ndarray):
        format = type(a).__name__.lower()
    elif format is None:
        format = "coo"
    if hasattr(a, "compressed_axes") and kwargs.get("compressed_axes") is None and format == "gcxs" and shape is None:
        kwargs["compressed_axes"] = a.compressed_axes
    return full(a.shape if shape is None else shape, fill_value, dtype=(a.dtype if dtype is None else dtype),
                format=format, device=device if device is not None else getattr(a, "device", None), **kwargs)


def zeros(shape, dtype=float, format="coo", *, device=None, **kwargs):
    return full(shape, 0, dtype=np.dtype(dtype), format=format, device=device, **kwargs)


def ones(shape, dtype=float, format="coo", *, device=None, **kwargs):
    return full(shape, 1, dtype=np.dtype(dtype), format=format, device=device, **kwargs)


def empty(shape, dtype=float, format="coo", *, device=None, **kwargs):
    return full(shape, 0, dtype=np.dtype(dtype), format=format, device=device, **kwargs)


def zeros_like(a, dtype=None, shape=None, format=None, *, device=None, **kwargs):
    return full_like(a, 0, dtype=dtype, shape=shape, format=format, device=device, **kwargs)


def ones_like(a, dtype=None, shape=None, format=None, *, device=None, **kwargs):
    return full_like(a, 1, dtype=dtype, shape=shape, format=format, device=device, **kwargs)


def empty_like(a, dtype=None, shape=None, format=None, *, device=None, **kwargs):
    return full_like(a, 0, dtype=dtype, shape=shape, format=format, device=device, **kwargs)


def eye(N, M=None, k=0, dtype=float, format="coo", *, device=None, **kwargs):
    """Ones on the k-th diagonal of an N x M array (reference `eye`, _common.py:1561-1625); coordinates are
    generated on the device."""
    N, M, k = int(N), int(N if M is None else M), int(k)
    length = builtins.min(N, M)
    if k > 0:
        length = builtins.max(builtins.min(length, M - k), 0)
    elif k < 0:
        length = builtins.max(builtins.min(length, N + k), 0)
    if length == 0:
        return zeros((N, M), dtype=dtype, format=format, device=device, **kwargs)
    d = torch.device(device) if device is not None else dev.default_device()
    rows = torch.arange(length, dtype=torch.int64, device=d) + builtins.max(-k, 0)  # memory plumbing: an index ramp
    coords = torch.stack([rows, rows + k])
    data = torch.ones(length, dtype=dev.torch_dtype(np.dtype(dtype)), device=d)
    out = COO(coords, data, shape=(N, M), has_duplicates=False, sorted=True)
    return out.asformat(format, **kwargs) if format != "coo" else out


# ---- selection and NaN-skipping reductions (reference _coo/common.py:334-733) ------------------------------------------
def where(condition, x=None, y=None):
    """`np.where` for sparse operands; with one argument, the coordinates of the stored elements (reference
    _coo/common.py:534-581)."""
    from ._umath import elemwise
    from ._utils import check_zero_fill_value

    if x is None and y is None:
        check_zero_fill_value(condition)
        return tuple(as_coo(condition).coords)
    if (x is None) != (y is None):
        raise ValueError("either both or neither of x and y should be given")
    return elemwise(np.where, condition, x, y)


def nonzero(x, /):
    """Coordinates of the stored elements, one array per dimension (reference `nonzero`)."""
    from ._utils import check_zero_fill_value

    check_zero_fill_value(x)
    return tuple(as_coo(x).coords)


def argwhere(a):
    """Stored coordinates, one row per element (reference `argwhere`, _coo/common.py:584-611)."""
    return torch.stack(list(where(a)), dim=1) if a.ndim else torch.zeros((a.nnz, 0), dtype=torch.int64)


def _replace_nan(x, value):
    if np.dtype(x.dtype).kind != "f":
        return x
    if isinstance(x, COO) and x.ndim and not np.isnan(x.fill_value) and x.data.dtype in (torch.float32, torch.float64):
        # where(isnan(x), value, x) leaves the fill value and every stored element that is no NaN as they are: nothing to do
        # without a NaN (one streaming read of the values instead of isnan's array + a three-operand merge: 0.8 ms at 10^7
        # stored elements), else a select on the value array under the same keys (results equal to the fill value pruned,
        # as `where` prunes them)
        if not x.nnz or not K.has_nan(x.data):
            return x
        from ._umath import unary_array

        data = x.data.contiguous()
        out = torch.empty_like(data)
        val = torch.tensor([value], dtype=data.dtype, device=data.device)
        _ffi.call("spamd_ewise_select", data.element_size(), int(data.numel()), dev.ptr(unary_array("isnan", data).view(torch.uint8)),
                  dev.ptr(val), 1, dev.ptr(data), 0, dev.ptr(out), dev.stream_ptr(data.device))
        return COO._from_sorted_keys(x.linear_loc(), out, x.shape, x.fill_value, x._index_dtype, prune=True)
    return where(np.isnan(x), value, x)


def nanreduce(x, method, identity=None, axis=None, keepdims=False, **kwargs):
    """NaN-skipping reduction: NaNs are replaced by the identity of `method`, then `reduce` (reference
    `nanreduce`, _coo/common.py:696-732)."""
    arr = _replace_nan(as_coo(x), method.identity if identity is None else identity)
    return arr.reduce(method, axis, keepdims, **kwargs)


def nansum(x, axis=None, keepdims=False, dtype=None, out=None):
    assert out is None
    return nanreduce(as_coo(x), np.add, axis=axis, keepdims=keepdims, dtype=dtype)


def nanprod(x, axis=None, keepdims=False, dtype=None, out=None):
    assert out is None
    return nanreduce(as_coo(x), np.multiply, axis=axis, keepdims=keepdims, dtype=dtype)


def _contains_nan(ar):
    if isinstance(ar, SparseArray):
        if np.dtype(ar.dtype).kind != "f":
            return False
        if ar.nnz != ar.size and np.isnan(ar.fill_value):
            return True
        return bool(K.has_nan(ar.data)) if ar.nnz else False
    return bool(np.isnan(ar))


def _nan_extreme(x, method, axis, keepdims, dtype):
    import warnings

    ar = as_coo(x).reduce(method, axis=axis, keepdims=keepdims, dtype=dtype)
    if _contains_nan(ar):
        warnings.warn("All-NaN slice encountered", RuntimeWarning, stacklevel=2)
    return ar


def nanmax(x, axis=None, keepdims=False, dtype=None, out=None):
    """Maximum skipping NaNs = reduce with np.fmax (reference `nanmax`, _coo/common.py:431-464)."""
    assert out is None
    return _nan_extreme(x, np.fmax, axis, keepdims, dtype)


def nanmin(x, axis=None, keepdims=False, dtype=None, out=None):
    assert out is None
    return _nan_extreme(x, np.fmin, axis, keepdims, dtype)


def nanmean(x, axis=None, keepdims=False, dtype=None, out=None):
    """Mean over the non-NaN elements (reference `nanmean`, _coo/common.py:364-415: same composition)."""
    import warnings

    assert out is None
    x = as_coo(x)
    if np.dtype(x.dtype).kind != "f":
        return x.mean(axis=axis, keepdims=keepdims, dtype=dtype)
    mask = np.isnan(x)
    x2 = _replace_nan(x, 0)
    nancount = mask.sum(axis=axis, dtype="i8", keepdims=keepdims)
    if axis is None:
        axis = tuple(range(x.ndim))
    elif not isinstance(axis, tuple):
        axis = (axis,)
    den = 1
    for i in axis:
        den *= x.shape[i]
    den = den - nancount
    if (den == 0).any():
        warnings.warn("Mean of empty slice", RuntimeWarning, stacklevel=1)
    num = x2.sum(axis=axis, dtype=dtype, keepdims=keepdims)
    with np.errstate(invalid="ignore", divide="ignore"):
        if num.ndim:
            return np.true_divide(num, den, casting="unsafe")
        return (num / den).astype(dtype if dtype is not None else x.dtype)


# ---- shape helpers of the array-API surface ---------------------------------------------------------------------------
def moveaxis(a, source, destination):
    src = [s % a.ndim for s in ((source,) if isinstance(source, int) else source)]
    dst = [d % a.ndim for d in ((destination,) if isinstance(destination, int) else destination)]
    if len(src) != len(dst):
        raise ValueError("`source` and `destination` arguments must have the same number of elements")
    order = [n for n in range(a.ndim) if n not in src]
    for d, s in sorted(zip(dst, src)):
        order.insert(d, s)
    return a.transpose(order)


def expand_dims(x, /, *, axis=0):
    """`_coo/common.py:1074-1133`: always a COO (the reference converts its input first)."""
    from ._array_api import _validate_coo_input

    x = _validate_coo_input(x)
    axes = (axis,) if isinstance(axis, int) else tuple(axis)
    nd = x.ndim + len(axes)
    axes = sorted(a % nd for a in axes)
    shape = list(x.shape)
    for a in axes:
        shape.insert(a, 1)
    return x.reshape(tuple(shape))


def squeeze(x, /, axis=None):
    if axis is None:
        axes = tuple(i for i, s in enumerate(x.shape) if s == 1)
    else:
        axes = tuple(a % x.ndim for a in ((axis,) if isinstance(axis, int) else axis))
        if builtins.any(x.shape[a] != 1 for a in axes):
            raise ValueError("cannot select an axis to squeeze out which has size not equal to one")
    return x.reshape(tuple(s for i, s in enumerate(x.shape) if i not in axes))
