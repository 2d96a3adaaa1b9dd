"""`DOK`: the dictionary-of-keys container (reference `sparse/numba_backend/_dok.py`).

DOK is the reference's MUTABLE format - a host dictionary `{(i, j, ...): value}` that is filled element by element or
slice by slice and then converted.  It is a host object there and it is one here: the dictionary never holds arithmetic.
Everything that computes goes through `to_coo()` (one packed upload, `COO.from_iter`) to the device formats; results of
elementwise operations on DOK operands come back as DOK, as in the reference (`_umath.py:419-420`)."""
import copy as _copy
from collections.abc import Iterable
from numbers import Integral

import numpy as np
from numpy.lib.mixins import NDArrayOperatorsMixin

from ._sparse_array import SparseArray
from ._utils import equivalent


class DOK(SparseArray, NDArrayOperatorsMixin):
    """N-D sparse array as `{coordinate tuple: value}` (reference `_dok.py:13-132`).

    `DOK(shape, data=None, dtype=None, fill_value=None)`; `shape` may also be a COO, a NumPy array or a SciPy sparse
    matrix to convert."""

    def __init__(self, shape, data=None, dtype=None, fill_value=None):
        from ._coo import COO, _is_scipy_sparse

        self.data = {}
        for kind, conv in ((COO, DOK.from_coo), (np.ndarray, DOK.from_numpy)):
            if isinstance(shape, kind):
                self._make_shallow_copy_of(conv(shape))
                return
        if _is_scipy_sparse(shape):
            self._make_shallow_copy_of(DOK.from_scipy_sparse(shape))
            return
        self._dtype = np.dtype(dtype)
        if not data:
            data = {}
        super().__init__(shape, fill_value=fill_value)
        if not isinstance(data, dict):
            raise ValueError("data must be a dict.")
        if not dtype:
            self._dtype = np.dtype("float64") if not len(data) else np.result_type(*(np.asarray(x).dtype for x in data.values()))
        for c, d in data.items():
            self[c] = d

    def _make_shallow_copy_of(self, other):
        self.data = other.data
        self._dtype = other._dtype
        super().__init__(other.shape, fill_value=other.fill_value)

    # ---- properties (the base class reads them off device tensors; a dictionary has none) ------------------------
    @property
    def dtype(self):
        return self._dtype

    @dtype.setter
    def dtype(self, value):
        self._dtype = np.dtype(value)

    @property
    def device(self):
        from . import _device

        return _device.default_device()

    @property
    def nnz(self):
        return len(self.data)

    @property
    def format(self):
        return "dok"

    @property
    def nbytes(self):
        return self.nnz * self.dtype.itemsize

    # ---- conversions ------------------------------------------------------------------------------------------------
    @classmethod
    def from_scipy_sparse(cls, x, /, *, fill_value=None):
        from ._coo import COO

        return COO.from_scipy_sparse(x, fill_value=fill_value).asformat(cls)

    @classmethod
    def from_coo(cls, x):
        ar = cls(x.shape, dtype=x.dtype, fill_value=x.fill_value)
        coords = x.coords.cpu().numpy().T
        data = x.data.cpu().numpy()
        for c, d in zip(coords, data, strict=True):
            ar.data[tuple(int(v) for v in c)] = d
        return ar

    def to_coo(self):
        return self.asformat("coo")

    @classmethod
    def from_numpy(cls, x):
        ar = cls(x.shape, dtype=x.dtype)
        coords = np.nonzero(x)
        for d, *c in zip(x[coords], *coords, strict=True):
            ar.data[tuple(int(v) for v in c)] = d
        return ar

    def todense(self):
        result = np.full(self.shape, self.fill_value, self.dtype)
        for c, d in self.data.items():
            result[c] = d
        return result

    def asformat(self, format, **kwargs):  # noqa: A002
        from ._utils import convert_format

        format = convert_format(format)  # noqa: A001
        if format == "dok":
            return self
        if format == "coo":
            from ._coo import COO

            if len(kwargs) != 0:
                raise ValueError(f"Extra kwargs found: {kwargs}")
            return COO.from_iter(self.data, shape=self.shape, fill_value=self.fill_value, dtype=self.dtype)
        return self.asformat("coo").asformat(format, **kwargs)

    def reshape(self, shape, order="C"):
        if order not in {"C", None}:
            raise NotImplementedError("The 'order' parameter is not supported")
        return DOK.from_coo(self.to_coo().reshape(shape))

    def copy(self, deep=True):
        return _copy.deepcopy(self) if deep else _copy.copy(self)

    def __str__(self):
        return f"<DOK: shape={self.shape!s}, dtype={self.dtype!s}, nnz={self.nnz:d}, fill_value={self.fill_value!s}>"

    __repr__ = __str__

    # ---- element access -----------------------------------------------------------------------------------------------
    def _index_sequences(self, key):
        if len(key) != self.ndim:
            raise NotImplementedError(f"Index sequences for all {self.ndim} array dimensions needed!")
        if not all(len(key[0]) == len(k) for k in key):
            raise IndexError("Unequal length of index sequences!")

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if all(isinstance(k, Iterable) for k in key):     # one sequence per dimension: the listed points, as a 1-D DOK
            self._index_sequences(key)
            found = {}
            for i, k in enumerate(zip(*key, strict=True)):
                k = tuple(int(v) for v in k)
                if k in self.data:
                    found[i] = self.data[k]
            return DOK(shape=len(key[0]), data=found, dtype=self.dtype, fill_value=self.fill_value)
        ret = self.asformat("coo")[key]
        if isinstance(ret, SparseArray):
            ret = ret.asformat("dok")
        return ret

    def __setitem__(self, key, value):
        value = np.asarray(value, dtype=self.dtype)
        if self.ndim == 1 and isinstance(key, Iterable) and all(isinstance(i, (int, np.integer)) for i in key):
            key = (key,)
        if isinstance(key, tuple) and all(isinstance(k, Iterable) for k in key):
            self._index_sequences(key)
            self._set_points(key, value)
            return
        self._set_block(self._normalise_key(key), value)

    def _normalise_key(self, key):
        """integers (negative ones wrapped, bounds checked) and slices, padded with full slices; `Ellipsis` expanded"""
        if not isinstance(key, tuple):
            key = (key,)
        if sum(1 for k in key if k is Ellipsis) > 1:
            raise IndexError("an index can only have a single ellipsis ('...')")
        real = sum(1 for k in key if k is not Ellipsis)
        if real > self.ndim:
            raise IndexError("Too many indices for array")
        rest = [slice(None)] * (self.ndim - real)
        at = next((i for i, k in enumerate(key) if k is Ellipsis), None)
        key = list(key) + rest if at is None else list(key[:at]) + rest + list(key[at + 1:])
        out = []
        for k, n in zip(key, self.shape, strict=True):
            if isinstance(k, Integral):
                i = int(k) + (n if k < 0 else 0)
                if not 0 <= i < n:
                    raise IndexError(f"Index {int(k)} is out of bounds for axis with size {n}")
                out.append(i)
            elif isinstance(k, slice):
                out.append(slice(*k.indices(n)))
            else:
                raise IndexError("All indices must be slices or integers when setting an item.")
        return out

    def _set_points(self, idxs, values):
        idxs = tuple(np.asanyarray(i) for i in idxs)
        if not all(np.issubdtype(k.dtype, np.integer) for k in idxs):
            raise IndexError("Indices must be sequences of integer types!")
        if idxs[0].ndim != 1:
            raise IndexError("Indices are not 1d sequences!")
        if values.ndim == 0:
            values = np.full(idxs[0].size, values, self.dtype)
        elif values.ndim > 1:
            raise ValueError(f"Dimension of values ({values.ndim}) must be 0 or 1!")
        if not idxs[0].shape == values.shape:
            raise ValueError(f"Shape mismatch of indices ({idxs[0].shape}) and values ({values.shape})!")
        for idx, value in zip(zip(*idxs, strict=True), values, strict=True):
            idx = tuple(int(v) for v in idx)
            if value != self.fill_value:
                self.data[idx] = value
            elif idx in self.data:
                del self.data[idx]

    def _set_block(self, key, value):
        """`self[key] = value` for integers and (normalised) slices: the value broadcasts NumPy-style from the right over
        the slice axes; elements equal to the fill value are removed, not stored (reference `_setitem`, `_dok.py:396-434`)."""
        n_slices = sum(1 for k in key if isinstance(k, slice))
        if n_slices - value.ndim < 0:
            raise ValueError("setting an array element with a sequence.")
        for i, ind in enumerate(key):
            if isinstance(ind, slice):
                positions = range(ind.start, ind.stop, ind.step)
                missing = n_slices - value.ndim > 0
                for v_idx, ki in enumerate(positions):
                    sub = value if missing else (value[0] if value.shape[0] == 1 else value[v_idx])
                    self._set_block(key[:i] + [ki] + key[i + 1:], sub)
                return
        where = tuple(key)
        if not equivalent(value, self.fill_value):
            self.data[where] = value[()]
        elif where in self.data:
            del self.data[where]
