"""`GCXS`: N-D compressed sparse array whose (data, indices, indptr) live in HBM.

Same constructor, attributes and methods as the reference container
(sparse/numba_backend/_compressed/compressed.py:80-187): the array is the CSR of the 2-D
matrix (prod(compressed dims) x prod(other dims)) obtained after ordering the axes as
`compressed_axes + remaining`.  The three arrays are torch tensors on the HIP device.
"""
import copy as _copy
from collections.abc import Iterable

import numpy as np
import torch

from . import _device as dev
from ._sparse_array import NDArrayOperatorsMixin, SparseArray
from ._utils import check_compressed_axes, normalize_axis, prod, zero_of_dtype


def _is_scipy_sparse(x):
    return hasattr(x, "tocsr") and hasattr(x, "format") and type(x).__module__.startswith("scipy.sparse")


def unified_index_dtype(indices_dtype, stored):
    """The one index width of a GCXS whose `indices` and `indptr` arrive with different ones: the indices' width, unless
    pointers of that width could not hold the number of stored elements."""
    return torch.int64 if indices_dtype == torch.int32 and stored > 2 ** 31 - 1 else indices_dtype


def _compressed_axis_slice(x, index):
    """`x[a:b]`, `x[i]` of a CSR matrix and `x[:, a:b]`, `x[:, j]` of a CSC one: a range of the pointers and the elements
    between its two ends (reference `_compressed/indexing.py:14-176` with only the compressed axis indexed, where
    `get_slicing_selection` copies whole rows) - two pointer values read, the slice copied, no conversion to COO and back
    (0.9-1.2 ms for 10^7 stored elements whatever the slice).  The result keeps the compressed axis, as the reference's
    does.  None = another form of index."""
    if x.ndim != 2 or x.compressed_axes not in ((0,), (1,)):
        return None
    idx = index if isinstance(index, tuple) else (index,)
    if not 1 <= len(idx) <= 2 or any(isinstance(i, (bool, np.bool_)) or not isinstance(i, (int, np.integer, slice)) for i in idx):
        return None
    idx = idx + (slice(None),) * (2 - len(idx))
    ca = x.compressed_axes[0]
    sel, other = idx[ca], idx[1 - ca]
    if not isinstance(other, slice) or other.indices(x.shape[1 - ca]) != (0, x.shape[1 - ca], 1):
        return None
    n = x.shape[ca]
    one = not isinstance(sel, slice)
    if one:
        a = int(sel) + (n if sel < 0 else 0)
        if not 0 <= a < n:
            raise IndexError(f"index {int(sel)} is out of bounds for axis {ca} with size {n}")
        b = a + 1
    else:
        a, b, step = sel.indices(n)
        if step != 1:
            return None
        b = max(a, b)
    p0, p1 = (int(v) for v in x.indptr[a:b + 1:b - a].tolist()) if b > a else (0, 0)
    data, indices = x.data[p0:p1].clone(), x.indices[p0:p1].clone()       # (copies: a view would start off the 16-byte boundary the kernels load from)
    if one:
        from ._coo import COO

        return COO(indices[None, :], data, shape=(x.shape[1 - ca],), has_duplicates=False, fill_value=x.fill_value).asformat("gcxs")
    indptr = x.indptr[a:b + 1].clone()
    if p0:
        from ._umath import binary_arrays

        indptr = binary_arrays("subtract", indptr, torch.tensor([p0], dtype=indptr.dtype, device=indptr.device), b_scalar=True)
    shape = (b - a, x.shape[1]) if ca == 0 else (x.shape[0], b - a)
    return GCXS((data, indices, indptr), shape=shape, compressed_axes=(ca,), fill_value=x.fill_value)


def _uncompressed_axis_slice(x, index):
    """`x[:, a:b]` of a CSR matrix, `x[a:b]` of a CSC one: the stored elements whose index lies in [a, b), compacted under a
    scan of that test; the new pointers are the scan read at the old ones (reference `_compressed/indexing.py:14-176`,
    `get_slicing_selection`).  Seven streaming kernels over the stored elements instead of a conversion to COO and back
    (1.17 ms at 10^7 stored elements).  None = another form of index."""
    from . import _kernels as K
    from ._umath import binary_arrays

    if x.ndim != 2 or x.compressed_axes not in ((0,), (1,)):
        return None
    idx = index if isinstance(index, tuple) else (index,)
    if not 1 <= len(idx) <= 2 or any(not isinstance(i, slice) for i in idx):
        return None
    idx = idx + (slice(None),) * (2 - len(idx))
    ca = x.compressed_axes[0]
    if idx[ca].indices(x.shape[ca]) != (0, x.shape[ca], 1):
        return None
    n = x.shape[1 - ca]
    a, b, step = idx[1 - ca].indices(n)
    if step != 1:
        return None
    b = max(a, b)
    if (a, b) == (0, n):
        return x
    shape = (x.shape[0], b - a) if ca == 0 else (b - a, x.shape[1])
    if x.nnz == 0 or b == a:
        return GCXS((x.data[:0].clone(), x.indices[:0].clone(), torch.zeros_like(x.indptr)), shape=shape, compressed_axes=(ca,),
                    fill_value=x.fill_value)
    ind = x.indices.contiguous()
    sc = lambda v: torch.tensor([v], dtype=ind.dtype, device=ind.device)
    inside = binary_arrays("logical_and", binary_arrays("greater_equal", ind, sc(a), b_scalar=True, out_bool_as=torch.uint8),
                           binary_arrays("less", ind, sc(b), b_scalar=True, out_bool_as=torch.uint8), out_bool_as=torch.uint8)
    flags = K.flag_ne_bits(inside, np.uint8(0))
    offs = K.exclusive_scan(flags)
    cnt = int(offs[-1])
    data, kept = K.compact(x.data, flags, offs, cnt), K.compact(ind, flags, offs, cnt)
    if a and cnt:
        kept = binary_arrays("subtract", kept, sc(a), b_scalar=True)
    indptr = K.gather(offs, x.indptr.to(torch.int64)).to(x.indptr.dtype)
    return GCXS((data, kept, indptr), shape=shape, compressed_axes=(ca,), fill_value=x.fill_value)


class GCXS(SparseArray, NDArrayOperatorsMixin):
    """Generalised compressed row/column storage on the device.

    Parameters follow the reference (`compressed.py:135-143`): `arg` is a
    `(data, indices, indptr)` triple (ndarrays or device tensors), a `COO`, another `GCXS`, a
    dense ndarray, or a SciPy sparse matrix.
    """

    __array_priority__ = 12

    def __init__(self, arg, shape=None, compressed_axes=None, prune=False, fill_value=None,
                 idx_dtype=None, device=None):
        from ._coo import COO
        from ._convert import coo_to_gcxs_arrays

        if _is_scipy_sparse(arg):
            arg = GCXS.from_scipy_sparse(arg, device=device)
        if isinstance(arg, (np.ndarray, torch.Tensor)) and not isinstance(arg, tuple):
            arg = COO.from_numpy(arg, fill_value=fill_value, device=device)
            arg, shape, compressed_axes, fill_value = coo_to_gcxs_arrays(arg, compressed_axes)
        elif isinstance(arg, COO):
            arg, shape, compressed_axes, fill_value = coo_to_gcxs_arrays(arg, compressed_axes, idx_dtype)
        elif isinstance(arg, GCXS):
            if compressed_axes is not None and arg.compressed_axes != compressed_axes:
                arg = arg.change_compressed_axes(compressed_axes)
            arg, shape, compressed_axes, fill_value = (
                (arg.data, arg.indices, arg.indptr), arg.shape, arg.compressed_axes, arg.fill_value)

        if shape is None:
            raise ValueError("missing `shape` argument")
        shape = tuple(int(s) for s in (shape if isinstance(shape, Iterable) else (shape,)))
        check_compressed_axes(len(shape), compressed_axes)
        if len(shape) == 1:
            compressed_axes = None

        data, indices, indptr = arg
        if device is None:
            device = next((t.device for t in (data, indices, indptr) if isinstance(t, torch.Tensor)), None)
        if device is None:
            device = dev.default_device()
        self.data = dev.to_device(data, device)
        self.indices = dev.to_device(indices, device)
        if isinstance(indptr, (list, tuple)) and len(indptr) == 0:
            indptr = np.empty(0, dtype=dev.np_dtype(self.indices))
        self.indptr = dev.to_device(indptr, device)
        if self.indices.dtype not in (torch.int32, torch.int64):
            self.indices = self.indices.to(torch.int64)
        if self.indptr.dtype != self.indices.dtype:
            # one index width for both arrays (the kernels take one): the indices' - unless the pointers then no longer hold
            # the number of stored elements (round 6: int32 indices with int64 pointers of 2.25 x 10^9 elements were narrowed
            # to int32 pointers here, and the first kernel that followed them faulted)
            width = unified_index_dtype(self.indices.dtype, int(self.data.numel()))
            if self.indices.dtype != width:
                self.indices = self.indices.to(width)
            self.indptr = self.indptr.to(width)
        if self.data.dim() != 1:
            raise ValueError("data must be a scalar or 1-dimensional.")

        self.shape = shape
        if fill_value is None:
            fill_value = zero_of_dtype(self.dtype)
        self._compressed_axes = tuple(int(c) for c in compressed_axes) if isinstance(compressed_axes, Iterable) else None
        self.fill_value = self.dtype.type(fill_value)
        if prune:
            self._prune()

    # ---- construction ------------------------------------------------------------------
    def copy(self, deep=True):
        if not deep:
            return _copy.copy(self)
        return GCXS((self.data.clone(), self.indices.clone(), self.indptr.clone()), shape=self.shape,
                    compressed_axes=self.compressed_axes, fill_value=self.fill_value)

    @classmethod
    def from_iter(cls, x, shape=None, compressed_axes=None, fill_value=None, idx_dtype=None, device=None):
        """GCXS from the iterable forms `COO.from_iter` accepts (reference compressed.py `from_iter`)."""
        from ._coo import COO

        return cls.from_coo(COO.from_iter(x, shape, fill_value, device=device), compressed_axes, idx_dtype)

    # pickling: host arrays only, derived layouts (CSR twin, tiled SpMM layout, NaN verdict) are not part of the state
    def __getstate__(self):
        return (self.data.cpu().numpy(), self.indices.cpu().numpy(), self.indptr.cpu().numpy(), self.shape,
                self._compressed_axes, self.fill_value)

    def __setstate__(self, state):
        data, indices, indptr, shape, compressed_axes, fill_value = state
        d = dev.default_device()
        self.data, self.indices, self.indptr = (dev.to_device(a, d) for a in (data, indices, indptr))
        self.shape, self._compressed_axes, self.fill_value = tuple(shape), compressed_axes, fill_value

    @classmethod
    def from_numpy(cls, x, compressed_axes=None, fill_value=None, idx_dtype=None, device=None):
        from ._coo import COO

        coo = COO.from_numpy(x, fill_value=fill_value, idx_dtype=idx_dtype, device=device)
        return cls.from_coo(coo, compressed_axes, idx_dtype)

    @classmethod
    def from_coo(cls, x, compressed_axes=None, idx_dtype=None):
        from ._convert import coo_to_gcxs_arrays

        arg, shape, compressed_axes, fill_value = coo_to_gcxs_arrays(x, compressed_axes, idx_dtype)
        return cls(arg, shape=shape, compressed_axes=compressed_axes, fill_value=fill_value)

    @classmethod
    def from_scipy_sparse(cls, x, /, *, fill_value=None, device=None):
        """CSR/CSC SciPy matrix -> GCXS with compressed_axes (0,)/(1,) (reference compressed.py:210-219)."""
        is_csc = x.format == "csc"
        ca = (1,) if is_csc else (0,)
        if not is_csc:
            x = x.asformat("csr")
        if not x.has_canonical_format:
            x.eliminate_zeros()
            x.sum_duplicates()
        return cls((x.data, x.indices, x.indptr), shape=x.shape, compressed_axes=ca,
                   fill_value=fill_value, device=device)

    # ---- properties ----------------------------------------------------------------------
    @property
    def compressed_axes(self):
        return self._compressed_axes

    @property
    def nnz(self):
        return int(self.data.shape[0])

    @property
    def format(self):
        return "gcxs"

    @property
    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.data, self.indices, self.indptr))

    @property
    def _axis_order(self):
        order = list(self.compressed_axes)
        order.extend(a for a in range(self.ndim) if a not in self.compressed_axes)
        return order

    @property
    def _axisptr(self):
        return len(self.compressed_axes)

    @property
    def _reordered_shape(self):
        return tuple(self.shape[i] for i in self._axis_order)

    @property
    def _compressed_shape(self):
        rs = self._reordered_shape
        return (prod(rs[: self._axisptr]), prod(rs[self._axisptr:]))

    @property
    def T(self):
        return self.transpose()

    @property
    def mT(self):
        if self.ndim < 2:
            raise ValueError("Cannot compute matrix transpose if `ndim < 2`.")
        axis = list(range(self.ndim))
        axis[-1], axis[-2] = axis[-2], axis[-1]
        return self.transpose(axis)

    def __str__(self):
        return (f"<GCXS: shape={self.shape}, dtype={self.dtype}, nnz={self.nnz}, fill_value={self.fill_value}, "
                f"compressed_axes={self.compressed_axes}, device={self.device}>")

    __repr__ = __str__

    # ---- conversions (device kernels live in _convert.py) -----------------------------------
    def change_compressed_axes(self, new_compressed_axes):
        """Re-compress along other axes (reference compressed.py:388-423)."""
        from ._convert import gcxs_relayout

        if new_compressed_axes == self.compressed_axes:
            return self
        if self.ndim == 1:
            raise NotImplementedError("no axes to compress for 1d array")
        new_compressed_axes = tuple(normalize_axis(a, self.ndim) for a in new_compressed_axes)
        if new_compressed_axes == self.compressed_axes:
            return self
        if len(new_compressed_axes) >= len(self.shape):
            raise ValueError("cannot compress all axes")
        if len(set(new_compressed_axes)) != len(new_compressed_axes):
            raise ValueError("repeated axis in compressed_axes")
        arg = gcxs_relayout(self, self.shape, tuple(range(self.ndim)), new_compressed_axes)
        return GCXS(arg, shape=self.shape, compressed_axes=new_compressed_axes, fill_value=self.fill_value)

    def tocoo(self):
        from ._convert import gcxs_to_coo

        return gcxs_to_coo(self)

    def todense(self):
        """Dense host ndarray (reference compressed.py:462-482 returns an ndarray)."""
        from ._convert import gcxs_todense

        return dev.to_numpy(gcxs_todense(self))

    def todense_device(self):
        """Dense tensor that stays in HBM."""
        from ._convert import gcxs_todense

        return gcxs_todense(self)

    def asformat(self, format, **kwargs):
        from ._utils import convert_format

        format = convert_format(format)
        ca = kwargs.pop("compressed_axes", None)
        if format == "gcxs":
            if ca is None or tuple(ca) == self.compressed_axes:
                return self
            return self.change_compressed_axes(tuple(ca))
        if format == "coo":
            if kwargs or ca is not None:
                raise TypeError("unexpected keyword arguments for format 'coo'")
            return self.tocoo()
        raise NotImplementedError(f"format {format!r} is not available in the hip backend")

    def maybe_densify(self, max_size=1000, min_density=0.25):
        if self.size <= max_size or self.density >= min_density:
            return self.todense()
        raise ValueError("Operation would require converting large sparse array to dense")

    def flatten(self, order="C"):
        if order not in {"C", None}:
            raise NotImplementedError("The `order` parameter is not supported.")
        return self.reshape(-1)

    def reshape(self, shape, order="C", compressed_axes=None):
        """reference compressed.py:622-682"""
        from ._convert import gcxs_relayout

        shape = tuple(shape) if isinstance(shape, Iterable) else (shape,)
        if order not in {"C", None}:
            raise NotImplementedError("The 'order' parameter is not supported")
        if any(d == -1 for d in shape):
            extra = int(self.size / max(1, prod(d for d in shape if d != -1)))
            shape = tuple(d if d != -1 else extra for d in shape)
        shape = tuple(int(d) for d in shape)
        if self.shape == shape:
            return self
        if self.size != prod(shape):
            raise ValueError(f"cannot reshape array of size {self.size} into shape {shape}")
        if len(shape) == 0:
            return self.tocoo().reshape(shape).asformat("gcxs")
        if compressed_axes is None:
            compressed_axes = self.compressed_axes if len(shape) == self.ndim else (int(np.argmin(shape)),)
        if self.ndim == 1:
            return self.tocoo().reshape(shape).asformat("gcxs", compressed_axes=compressed_axes)
        if len(shape) == 1:
            return self.tocoo().reshape(shape).asformat("gcxs")
        check_compressed_axes(len(shape), compressed_axes)
        arg = gcxs_relayout(self, shape, tuple(range(self.ndim)), tuple(compressed_axes), reshape=True)
        return GCXS(arg, shape=shape, compressed_axes=tuple(compressed_axes), fill_value=self.fill_value)

    def transpose(self, axes=None, compressed_axes=None):
        """reference compressed.py:688-741; the 2-D case is metadata-only (:743-768)."""
        from ._convert import gcxs_relayout

        if axes is None:
            axes = list(reversed(range(self.ndim)))
        axes = normalize_axis(tuple(axes), self.ndim)
        if len(set(axes)) != len(axes):
            raise ValueError("repeated axis in transpose")
        if set(axes) != set(range(self.ndim)):
            raise ValueError("axes don't match array")
        axes = tuple(axes)
        if axes == tuple(range(self.ndim)):
            return self
        if self.ndim == 2:
            return self._2d_transpose()
        shape = tuple(self.shape[ax] for ax in axes)
        if compressed_axes is None:
            compressed_axes = (int(np.argmin(shape)),)
        check_compressed_axes(len(shape), compressed_axes)
        arg = gcxs_relayout(self, shape, axes, tuple(compressed_axes), transpose=True)
        return GCXS(arg, shape=shape, compressed_axes=tuple(compressed_axes), fill_value=self.fill_value)

    def _2d_transpose(self):
        ca = ((self.compressed_axes[0] + 1) % 2,)
        return GCXS((self.data, self.indices, self.indptr), shape=self.shape[::-1], compressed_axes=ca,
                    fill_value=self.fill_value)

    def __getitem__(self, index):
        """Only the forms N-D `matmul` needs: `x[(None,) * k]` and `x[i]`."""
        if isinstance(index, tuple) and all(i is None for i in index):
            return self.tocoo()[index].asformat("gcxs") if index else self
        fast = _compressed_axis_slice(self, index)
        if fast is None:
            fast = _uncompressed_axis_slice(self, index)
        if fast is not None:
            return fast
        if isinstance(index, (int, np.integer)) and self.ndim > 1:
            from ._batched import take_leading

            return take_leading(self, int(index))
        from ._indexing import getitem

        return getitem(self, index)

    def dot(self, other):
        from ._dot import dot

        return dot(self, other)

    def __matmul__(self, other):
        return _dot_module().matmul(self, other)

    def __rmatmul__(self, other):
        return _dot_module().matmul(other, self)

    def _prune(self):
        """Drop stored entries bit-equal to the fill value (reference compressed.py:816-848)."""
        from ._convert import gcxs_prune

        gcxs_prune(self)

    def _make_shallow_copy_of(self, other):
        self.data, self.indices, self.indptr = other.data, other.indices, other.indptr
        self.shape = other.shape
        self._compressed_axes = other.compressed_axes
        self.fill_value = other.fill_value
        from ._dot import drop_derived

        drop_derived(self)   # CSR twin / tiled block streams / NaN verdict were built from the old buffers

    def to_scipy_sparse(self, accept_fv=None):
        import scipy.sparse

        from ._utils import check_fill_value

        check_fill_value(self, accept_fv=accept_fv)
        if self.ndim != 2:
            raise ValueError("Can only convert a 2-dimensional array to a Scipy sparse matrix.")
        cls = scipy.sparse.csr_matrix if self.compressed_axes == (0,) else scipy.sparse.csc_matrix
        return cls((dev.to_numpy(self.data), dev.to_numpy(self.indices), dev.to_numpy(self.indptr)), shape=self.shape)


_DOT = None


def _dot_module():
    """`._dot`, imported at first use (it imports this module) and kept: a function-local `from ._dot import matmul` costs
    ~1 us per `a @ b`, which is visible in products of a few hundred stored elements"""
    global _DOT
    if _DOT is None:
        from . import _dot

        _DOT = _dot
    return _DOT
