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
    def __array_function__(self, func, types, args, kwargs):
        """Route `np.<func>(sparse, ...)` to the same-named function of this package
        (reference _sparse_array.py:282-308)."""
        import sparse_amd as module

        sparse_func = None
        try:
            submodules = getattr(func, "__module__", "numpy").split(".")[1:]
            for sub in submodules:
                module = getattr(module, sub)
            sparse_func = getattr(module, func.__name__)
        except AttributeError:
            pass
        else:
            return sparse_func(*args, **kwargs)
        try:
            sparse_func = getattr(type(self), func.__name__)
        except AttributeError:
            pass
        if not isinstance(sparse_func, property) and callable(sparse_func):
            return sparse_func(*args, **kwargs)
        if sparse_func is None:
            return NotImplemented
        return sparse_func.__get__(self)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        """`np.<ufunc>(...)` with a sparse operand: "__call__" -> elemwise, "reduce" -> reduce
        (reference _sparse_array.py:322-370)."""
        from ._umath import elemwise

        out = kwargs.pop("out", None)
        if out is not None and not all(isinstance(x, type(self)) for x in out):
            return NotImplemented
        if getattr(ufunc, "signature", None) is not None:
            return self.__array_function__(ufunc, (np.ndarray, type(self)), inputs, kwargs)
        if out is not None:
            kwargs["dtype"] = out[0].dtype
        if method == "outer":
            method = "__call__"
            cum_ndim = 0
            inputs_transformed = []
            for inp in inputs:
                inputs_transformed.append(inp[(Ellipsis,) + (None,) * cum_ndim])
                cum_ndim += inp.ndim
            inputs = tuple(inputs_transformed)
        if method == "__call__":
            result = elemwise(ufunc, *inputs, **kwargs)
        elif method == "reduce":
            result = SparseArray._reduce(ufunc, *inputs, **kwargs)
        else:
            return NotImplemented
        if out is not None:
            (out,) = out
            if out.shape != result.shape:
                raise ValueError(f"non-broadcastable output operand with shape {out.shape} "
                                 f"doesn't match the broadcast shape {result.shape}")
            out._make_shallow_copy_of(result)
            return out
        return result

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
        """sum / n in the reference's order of operations (_sparse_array.py:645-723)."""
        if axis is None:
            axis = tuple(range(self.ndim))
        elif not isinstance(axis, tuple):
            axis = (axis,)
        den = 1
        for ax in normalize_axis(axis, self.ndim):
            den *= self.shape[ax]
        if dtype is None:
            if issubclass(self.dtype.type, (np.integer, np.bool_)):
                dtype = inter_dtype = np.dtype("f8")
            else:
                dtype = self.dtype
                inter_dtype = np.dtype("f4") if issubclass(dtype.type, np.float16) else dtype
        else:
            inter_dtype = dtype
        num = self.sum(axis=axis, keepdims=keepdims, dtype=inter_dtype)
        if num.ndim:
            out_ = np.true_divide(num, den, casting="unsafe")
            return out_.astype(dtype) if out_.dtype != dtype else out_
        return np.divide(num, den, dtype=dtype, out=out)

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
        if out is not None and not isinstance(out, tuple):
            out = (out,)
        return self.__array_ufunc__(np.round, "__call__", self, decimals=decimals, out=out)

    def clip(self, min=None, max=None, out=None):
        if min is None and max is None:
            raise ValueError("One of max or min must be given.")
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
