"""`SparseArray`: the common base of the HIP-resident containers.

Mirrors the attribute/protocol surface of the reference's abstract base
(sparse/numba_backend/_sparse_array.py): shape/ndim/size/density/nnz/fill_value/device, the
NumPy protocols (`__array_ufunc__` -> `elemwise`, `__array_function__` -> same-named
function of this package), `reduce` and the reductions built on it.  The arrays themselves
live in HBM as torch tensors; arithmetic is done by libsparse_amd.so.
"""
from collections.abc import Iterable
from numbers import Integral

import numpy as np

from ._utils import normalize_axis, zero_of_dtype

# add -> multiply, multiply -> power: the closed form that folds the implicit fill values of a
# group into a sum / product (reference _sparse_array.py:14, used at :409-421)
_REDUCE_SUPER_UFUNC = {np.add: np.multiply, np.multiply: np.power}


class SparseArray:
    """Base class of `COO` and `GCXS` (reference numba_backend/_sparse_array.py:17)."""

    __array_priority__ = 12

    def __init__(self, shape, fill_value=None):
        if not isinstance(shape, Iterable):
            shape = (shape,)
        shape = tuple(shape)
        if not all(isinstance(s, Integral) and int(s) >= 0 for s in shape):
            raise ValueError("shape must be an non-negative integer or a tuple of non-negative integers.")
        self.shape = tuple(int(s) for s in shape)
        if fill_value is None:
            self.fill_value = zero_of_dtype(self.dtype)
        elif not hasattr(fill_value, "dtype") or fill_value.dtype != self.dtype:
            self.fill_value = self.dtype.type(fill_value)
        else:
            self.fill_value = fill_value

    # ---- basic properties -------------------------------------------------------------
    @property
    def dtype(self):
        """NumPy dtype of the stored values."""
        from ._device import np_dtype

        return np_dtype(self.data)

    @property
    def device(self):
        """The torch device the arrays live on (the reference hard-wires "cpu",
        _sparse_array.py:50-52)."""
        return self.data.device

    def to_device(self, device, /, *, stream=None):
        """Array-API device move (reference _sparse_array.py:54-59, which knows only "cpu" and returns `self`).  Here
        the stored arrays are HIP device tensors: the same device returns `self`, another HIP device returns a copy
        whose buffers live there (derived layouts are rebuilt on demand), anything else is refused with the
        reference's exception type."""
        import copy as _copy

        import torch

        from ._dot import drop_derived

        if stream is not None:
            raise ValueError("The stream argument to to_device() is not supported")
        d = torch.device(device) if not isinstance(device, torch.device) else device
        if d.type != "cuda":
            raise ValueError(f"Unsupported device {device!r}: sparse_amd arrays live in HIP device memory")
        cur = self.device
        if d.index is None:
            d = torch.device("cuda", torch.cuda.current_device() if cur.index is None else cur.index)
        if d == cur:
            return self
        out = _copy.copy(self)
        drop_derived(out)
        out.__dict__.pop("_sddmm_plan", None)
        for name in ("data", "coords", "indices", "indptr", "_keys"):
            t = getattr(self, name, None)
            if isinstance(t, torch.Tensor):
                setattr(out, name, t.to(d))
        if getattr(out, "_cache", None) is not None:
            out._cache = type(out._cache)()
        return out

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def density(self):
        return self.nnz / self.size if self.size else 0.0

    def __len__(self):
        if self.ndim == 0:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    # ---- densification guard (reference _sparse_array.py:272-280) ----------------------
    def __array__(self, *args, **kwargs):
        from ._settings import AUTO_DENSIFY

        if not AUTO_DENSIFY:
            raise RuntimeError(
                "Cannot convert a sparse array to dense automatically. To manually densify, use the todense method."
            )
        return np.asarray(self.todense(), *args, **kwargs)

    # ---- NumPy protocols ---------------------------------------------------------------
    # The two dispatch hooks below reproduce the observable behaviour of the reference's hooks
    # (sparse/numba_backend/_sparse_array.py:282-370, BSD-3-Clause, (c) the pydata/sparse developers): which package
    # function a NumPy function resolves to, which exceptions an illegal `out=` raises, how `outer` is expressed as a
    # broadcast call.  The bodies are this package's own.
    def __array_function__(self, func, types, args, kwargs):
        """`np.<func>(sparse, ...)` -> the function of the same dotted name in this package, else an attribute of the
        container class (a method is called with the original arguments, a property is read for the one-argument
        form `np.<name>(x)`), else `NotImplemented`."""
        import sparse_amd

        where = sparse_amd
        for part in getattr(func, "__module__", "numpy").split(".")[1:]:   # numpy.linalg.x -> sparse_amd.linalg.x
            where = getattr(where, part, None)
            if where is None:
                break
        target = getattr(where, func.__name__, None) if where is not None else None
        if target is not None:
            return target(*args, **kwargs)
        member = getattr(type(self), func.__name__, None)
        if member is None:
            return NotImplemented
        if callable(member):
            return member(*args, **kwargs)
        if len(args) == 1 and not kwargs:
            return getattr(self, func.__name__)
        return NotImplemented

    @staticmethod
    def _check_out_casting(ufunc, method, inputs, out, kwargs):
        """An `out=` whose dtype the ufunc may not cast into must fail BEFORE any work, with NumPy's own exception
        (`UFuncTypeError`): NumPy itself is asked, on one-element stand-ins of the operands' dtypes
        (reference _sparse_array.py:333-342)."""
        probe = [np.empty((1,), dtype=v.dtype) if hasattr(v, "dtype") else v for v in inputs]
        probe_out = tuple(np.empty((1,), dtype=o.dtype) for o in out)
        kw = dict(kwargs)
        if method == "reduce":
            kw["axis"] = None
        getattr(ufunc, method)(*probe, out=probe_out[0] if len(probe_out) == 1 else probe_out, **kw)

    @staticmethod
    def _outer_as_broadcast(inputs):
        """`ufunc.outer(a, b, ...)` as a broadcasting call: operand i keeps its own axes and gets one trailing unit
        axis for every axis of the operands AFTER it, so the result's axes are a's, then b's, ...
        (reference _sparse_array.py:343-352)."""
        trailing = 0
        shaped = []
        for v in reversed(inputs):
            shaped.append(v[(Ellipsis,) + (None,) * trailing])
            trailing += v.ndim
        return tuple(reversed(shaped))

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        """`np.<ufunc>(...)` with a sparse operand: "__call__" and "outer" -> elemwise, "reduce" -> reduce, generalised
        ufuncs (matmul, ...) -> `__array_function__`, anything else `NotImplemented` (reference
        _sparse_array.py:322-370)."""
        from ._umath import elemwise

        out = kwargs.pop("out", None)
        if out is not None and not all(isinstance(o, type(self)) for o in out):
            return NotImplemented
        if getattr(ufunc, "signature", None) is not None:
            return self.__array_function__(ufunc, (np.ndarray, type(self)), inputs, kwargs)
        if out is not None:
            SparseArray._check_out_casting(ufunc, method, inputs, out, kwargs)
            kwargs["dtype"] = out[0].dtype
        if method == "outer":
            inputs, method = SparseArray._outer_as_broadcast(inputs), "__call__"
        if method == "__call__":
            result = elemwise(ufunc, *inputs, **kwargs)
        elif method == "reduce":
            result = SparseArray._reduce(ufunc, *inputs, **kwargs)
        else:
            return NotImplemented
        if out is None:
            return result
        (target,) = out
        if target.shape != result.shape:
            raise ValueError(f"non-broadcastable output operand with shape {target.shape} "
                             f"doesn't match the broadcast shape {result.shape}")
        target._make_shallow_copy_of(result)
        return target

    @staticmethod
    def _reduce(method, *args, **kwargs):
        assert len(args) == 1
        self = args[0]
        if isinstance(self, np.ndarray):
            return method.reduce(self, **kwargs)
        return self.reduce(method, **kwargs)

    # ---- reductions (reference _sparse_array.py:372-437; Appendix D5) -------------------
    def reduce(self, method, axis=(0,), keepdims=False, **kwargs):
        """Reduce with a binary NumPy ufunc over `axis`, accounting for implicit fill values."""
        from ._reduce import reduce_impl

        return reduce_impl(self, method, axis=axis, keepdims=keepdims, **kwargs)

    def sum(self, axis=None, keepdims=False, dtype=None, out=None):
        return np.add.reduce(self, out=out, axis=axis, keepdims=keepdims, dtype=dtype)

    def prod(self, axis=None, keepdims=False, dtype=None, out=None):
        return np.multiply.reduce(self, out=out, axis=axis, keepdims=keepdims, dtype=dtype)

    def max(self, axis=None, keepdims=False, out=None):
        return np.maximum.reduce(self, out=out, axis=axis, keepdims=keepdims)

    amax = max

    def min(self, axis=None, keepdims=False, out=None):
        return np.minimum.reduce(self, out=out, axis=axis, keepdims=keepdims)

    amin = min

    def any(self, axis=None, keepdims=False, out=None):
        return np.logical_or.reduce(self, out=out, axis=axis, keepdims=keepdims)

    def all(self, axis=None, keepdims=False, out=None):
        return np.logical_and.reduce(self, out=out, axis=axis, keepdims=keepdims)

    def mean(self, axis=None, keepdims=False, dtype=None, out=None):
        """Arithmetic mean = sum in an intermediate dtype, divided by the number of reduced positions, cast to the
        result dtype — the dtype rules are NumPy's `mean` as the reference applies them (_sparse_array.py:645-723):
        integers and booleans average in float64, float16 sums in float32 and is cast back, an explicit `dtype` is
        used for both."""
        axes = tuple(range(self.ndim)) if axis is None else (axis if isinstance(axis, tuple) else (axis,))
        n_reduced = 1
        for ax in normalize_axis(axes, self.ndim):
            n_reduced *= self.shape[ax]
        if dtype is not None:
            result_dtype = sum_dtype = dtype
        elif self.dtype.kind in "iub":
            result_dtype = sum_dtype = np.dtype("f8")
        else:
            result_dtype = self.dtype
            sum_dtype = np.dtype("f4") if self.dtype == np.dtype("f2") else self.dtype
        total = self.sum(axis=axes, keepdims=keepdims, dtype=sum_dtype)
        if total.ndim == 0:
            return np.divide(total, n_reduced, dtype=result_dtype, out=out)
        quotient = np.true_divide(total, n_reduced, casting="unsafe")
        return quotient if quotient.dtype == result_dtype else quotient.astype(result_dtype)

    def var(self, axis=None, dtype=None, out=None, ddof=0, keepdims=False):
        """Variance (reference _sparse_array.py:725-814), evaluated per group on the device."""
        from ._reduce import var_impl

        return var_impl(self, axis=axis, dtype=dtype, ddof=ddof, keepdims=keepdims)

    def std(self, axis=None, dtype=None, out=None, ddof=0, keepdims=False):
        ret = self.var(axis=axis, dtype=dtype, out=out, ddof=ddof, keepdims=keepdims)
        return np.sqrt(ret)

    # ---- elementwise conveniences (all are `__array_ufunc__` calls in the reference) ----
    def astype(self, dtype, casting="unsafe", copy=True):
        """Copy with the values cast (reference _sparse_array.py:592-617: an elementwise call,
        so results equal to the cast fill value are pruned — Appendix C.12)."""
        if self.dtype == dtype and not copy:
            return self
        from ._umath import elemwise

        return elemwise(np.ndarray.astype, self, dtype=np.dtype(dtype), casting=casting)

    @property
    def real(self):
        return self.__array_ufunc__(np.real, "__call__", self)

    @property
    def imag(self):
        return self.__array_ufunc__(np.imag, "__call__", self)

    def conj(self):
        return np.conj(self)

    def round(self, decimals=0, out=None):
        if out is None and isinstance(decimals, (int, np.integer)) and not isinstance(decimals, (bool, np.bool_)) \
                and np.dtype(self.dtype) in (np.dtype("f4"), np.dtype("f8")) and abs(int(decimals)) <= 15:
            # NumPy's own recipe (PyArray_Round: multiply by 10^d, rint, divide - the other way round for d < 0) as three
            # device passes over the stored values; `np.round` as ONE function with a keyword has no kernel and was evaluated
            # on the host: 114 ms at 10^7 stored elements against 0.5 (tools/r06/bcast_sparse_sweep.py)
            d = int(decimals)
            if d == 0:
                return np.rint(self)
            m = float(10 ** abs(d))
            return np.rint(self * m) / m if d > 0 else np.rint(self / m) * m
        if out is not None and not isinstance(out, tuple):
            out = (out,)
        return self.__array_ufunc__(np.round, "__call__", self, decimals=decimals, out=out)

    round_ = round

    def clip(self, min=None, max=None, out=None):
        if min is None and max is None:
            raise ValueError("One of max or min must be given.")
        scalar = lambda v: v is None or (isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, (bool, np.bool_)))
        if out is None and scalar(min) and scalar(max) and np.dtype(self.dtype).kind in "fiu":
            # np.clip(a, lo, hi) is minimum(maximum(a, lo), hi) - NaNs of `a` pass through both -: two device ufuncs instead
            # of a host evaluation of the three-argument function (93 ms at 10^7 stored elements against 0.4)
            r = self if min is None else np.maximum(self, min)
            return r if max is None else np.minimum(r, max)
        if out is not None and not isinstance(out, tuple):
            out = (out,)
        return self.__array_ufunc__(np.clip, "__call__", self, a_min=min, a_max=max, out=out)

    def __array_namespace__(self, *, api_version=None):
        import sparse_amd

        return sparse_amd

    def __bool__(self):
        return self._to_scalar(bool)

    def __float__(self):
        return self._to_scalar(float)

    def __int__(self):
        return self._to_scalar(int)

    def __complex__(self):
        return self._to_scalar(complex)

    def __index__(self):
        return self._to_scalar(int)

    def _to_scalar(self, builtin):
        if self.size != 1 or self.shape != ():
            raise ValueError(f"{builtin} can be computed for one-element arrays only.")
        return builtin(self.todense().flatten()[0])

    def isinf(self):
        return np.isinf(self)

    def isnan(self):
        return np.isnan(self)


# binary/unary operators: the reference gets these from numpy.lib.mixins.NDArrayOperatorsMixin
# (core.py:26, compressed.py:80); the same mixin is used by COO and GCXS here.
from numpy.lib.mixins import NDArrayOperatorsMixin  # noqa: E402,F401
