"""`COO`: N-D coordinate-format sparse array whose (coords, data) live in HBM.

Same constructor contract as the reference container (sparse/numba_backend/_coo/core.py:
198-291): the canonical form is sorted by C-order linear index, duplicate-free and, on
request, pruned of stored fill values (Appendix D1).  All of that runs on the device on
64-bit linear keys (csrc/prims.hip); torch only owns the memory.
"""
import copy as _copy
import warnings
from collections.abc import Iterable

import numpy as np
import torch

from . import _device as dev
from . import _kernels as K
from ._sparse_array import NDArrayOperatorsMixin, SparseArray
from ._utils import can_store, normalize_axis, prod, zero_of_dtype


def _is_scipy_sparse(x):
    return hasattr(x, "tocoo") and hasattr(x, "format") and type(x).__module__.startswith("scipy.sparse")


def _index_tensor(coords, device, idx_dtype=None):
    t = dev.to_device(coords, device)
    if idx_dtype is not None:
        want = torch.int32 if np.dtype(idx_dtype).itemsize <= 4 and np.dtype(idx_dtype) != np.dtype("uint32") else torch.int64
        return t.to(want)
    if t.dtype == torch.int32 or t.dtype == torch.int64:
        return t
    if t.dtype in (torch.int8, torch.uint8, torch.int16):
        return t.to(torch.int32)  # narrow index dtypes are widened to int32 on the device
    return t.to(torch.int64)


SUM_RUNS_PROBE_MIN = 1 << 12      # fewer elements: even one run of all of them is short work

class COO(SparseArray, NDArrayOperatorsMixin):
    """Coordinate-format sparse array on the HIP device.

    Parameters follow the reference (`_coo/core.py:198-209`): `coords` is `[ndim, nnz]`,
    `data` is `[nnz]` (or a scalar), plus the `has_duplicates` / `sorted` / `prune` promises
    the producer makes about its output.
    """

    __array_priority__ = 12

    def __init__(self, coords, data=None, shape=None, has_duplicates=True, sorted=False, prune=False,
                 cache=False, fill_value=None, idx_dtype=None, device=None):
        self._cache = None
        if isinstance(coords, COO):
            self._make_shallow_copy_of(coords)
            if data is not None or shape is not None:
                raise ValueError("If `coords` is `COO`, then no other arguments should be provided.")
            if fill_value is not None:
                self.fill_value = self.dtype.type(fill_value)
            return
        if cache:
            self.enable_caching()
        if data is None:
            arr = as_coo(coords, shape=shape, fill_value=fill_value, idx_dtype=idx_dtype, device=device)
            self._make_shallow_copy_of(arr)
            if cache:
                self.enable_caching()
            return

        if device is None:
            device = next((t.device for t in (coords, data) if isinstance(t, torch.Tensor) and t.is_cuda), None)
        if device is None:
            device = dev.default_device()
        device = torch.device(device)
        self.data = dev.to_device(data, device)
        self.coords = _index_tensor(coords, device, idx_dtype)

        if self.coords.dim() == 1:
            if self.coords.numel() == 0 and shape is not None:
                nd = len(shape) if isinstance(shape, Iterable) else 1
                self.coords = self.coords.reshape((nd, int(self.data.numel()) if self.data.dim() else 0))
            else:
                self.coords = self.coords[None, :]
        if self.data.dim() == 0:
            self.data = self.data.expand(self.coords.shape[1]).contiguous()
        if self.data.dim() != 1:
            raise ValueError("`data` must be a scalar or 1-dimensional.")
        if shape is None:
            raise ValueError("`shape` was not provided.")
        if not isinstance(shape, Iterable):
            shape = (shape,)
        shape = tuple(int(s) for s in shape)
        if shape and not self.coords.numel():
            self.coords = torch.zeros((len(shape), 0), dtype=self.coords.dtype, device=device)
        super().__init__(shape, fill_value=fill_value)
        if idx_dtype and not can_store(idx_dtype, max(shape) if shape else 0):
            raise ValueError(f"cannot cast array with shape {shape} to dtype {idx_dtype}.")

        if self.shape:
            if int(self.data.shape[0]) != int(self.coords.shape[1]):
                raise ValueError("The data length does not match the coordinates given.\n"
                                 f"len(data) = {int(self.data.shape[0])}, but {int(self.coords.shape[1])} coords specified.")
            if len(self.shape) != int(self.coords.shape[0]):
                raise ValueError("Shape specified by `shape` doesn't match the shape of `coords`; "
                                 f"len(shape)={len(shape)} != coords.shape[0]={int(self.coords.shape[0])}"
                                 f"(and coords.shape={tuple(self.coords.shape)})")
        from ._settings import WARN_ON_TOO_DENSE

        if WARN_ON_TOO_DENSE and self.nbytes >= self.size * self.data.element_size():
            warnings.warn("Attempting to create a sparse array that takes no less memory than than an equivalent "
                          "dense array. You may want to use a dense array here instead.", RuntimeWarning, stacklevel=1)

        self._keys = None  # sorted C-order linear keys when known (device int64[nnz])
        self._coords_dtype = self.__dict__["_coords"].dtype
        if not sorted or has_duplicates:
            self._canonicalize(do_sort=not sorted, do_sum=has_duplicates)
        if prune:
            self._prune()

    # ---- coordinates: materialised lazily --------------------------------------------------------
    # Every kernel on the path works on the sorted C-order linear keys (`_keys`); the [ndim, nnz] coordinate
    # matrix the reference stores is only needed at the API boundary, so results built from keys
    # (`_from_sorted_keys`: elementwise, transpose, reshape, from_numpy) split them into coordinates on first
    # access of `.coords` (saves 24 B per element of writes for a 3-D result that is consumed by another op).
    @property
    def coords(self):
        c = self.__dict__.get("_coords")
        if c is None:
            c = K.delinearize(self._keys, self.shape, self._coords_dtype)
            self.__dict__["_coords"] = c
        return c

    @coords.setter
    def coords(self, value):
        self.__dict__["_coords"] = value

    @property
    def _index_dtype(self):
        c = self.__dict__.get("_coords")
        return c.dtype if c is not None else self._coords_dtype

    @classmethod
    def _from_sorted_keys(cls, keys, data, shape, fill_value, idx_dtype, prune=False):
        """Canonical COO from sorted, duplicate-free linear keys; coordinates are produced on demand."""
        out = cls.__new__(cls)
        out.__dict__["_coords"] = None
        out._coords_dtype = idx_dtype
        out._keys = keys
        out.data = data
        out.shape = tuple(int(d) for d in shape)
        out.fill_value = fill_value
        out._cache = None
        if prune:
            out._prune()
        return out

    # ---- canonical form (Appendix D1) ---------------------------------------------------------
    def linear_loc(self):
        """C-order linear index of every stored element (reference core.py `linear_loc`)."""
        if getattr(self, "_keys", None) is None or self._keys.numel() != self.nnz:
            self._keys = K.linearize(self.coords, self.shape)
        return self._keys

    def _canonicalize(self, do_sort, do_sum):
        """`_sort_indices` (stable, only if needed) + `_sum_duplicates` (reference
        core.py:1294-1353)."""
        if self.ndim == 0 or self.nnz == 0:
            if self.ndim == 0 and self.nnz > 1 and do_sum:
                # 0-d array: every stored element is a duplicate of the single position
                keys = torch.zeros(self.nnz, dtype=torch.int64, device=self.device)
                self._sum_runs(keys)
            return
        if not K.coords_in_range(self.coords, self.shape):
            # (the reference trusts its caller here; an out-of-range coordinate would turn into an out-of-bounds
            # scatter on the device, so this backend refuses it)
            raise IndexError(f"coords contain entries outside the array shape {self.shape}")
        keys = K.linearize(self.coords, self.shape)
        unsorted, dup = K.keys_check(keys)
        if do_sort and unsorted:
            keys, perm = K.sort_keys(keys, max(self.size - 1, 1))
            self.coords = K.gather(self.coords, perm)
            self.data = K.gather(self.data, perm)
            if do_sum:  # adjacent-equal test is only meaningful on sorted keys
                _, dup = K.keys_check(keys)
        elif unsorted and not do_sort:
            raise ValueError("COO was declared sorted=True but its coordinates are not sorted")
        if do_sum and dup:
            self._sum_runs(keys)
        else:
            self._keys = keys

    def _sum_runs(self, keys):
        """Keep the first coordinate of each run of equal keys and sum the run's data left to
        right in the data dtype (`np.add.reduceat`, reference core.py:1340-1353)."""
        from ._reduce import group_reduce, has_long_run, segment_reduce

        flags = K.flag_heads(keys)
        offs = K.exclusive_scan(flags)
        count = int(offs[-1])
        data = None
        if self.data.dtype in K._CODE_T and self.data.dtype != torch.bool and keys.numel() >= SUM_RUNS_PROBE_MIN \
                and self.data.is_contiguous() and self.data.data_ptr() % 16 == 0 and keys.data_ptr() % 16 == 0:
            # a run per thread (or per wave) is what a hot coordinate must not meet: 10^6 duplicates of ONE coordinate among
            # 10^6 others took 220 ms, 10^7 2.4 s (tools/r06/dup_hot.py).  When the scan of the head flags shows a window of
            # 1024 elements without a head (`has_long_run`: n / 1024 words read), the grouped reduce sums the runs - any
            # length, at streaming speed; there the reference's reduceat is pairwise, not left-to-right, so no order is "the"
            # order.  Otherwise the run-per-thread sums: left to right, the reference's bits for runs under 8.
            if has_long_run(offs):
                data = group_reduce(keys, 1, self.data, "add", key_bound=max(int(self.size), 1), sync=False)[1][:count]
        self.data = data if data is not None else segment_reduce(self.data, flags, offs, count, "add")
        self.coords = K.compact(self.coords, flags, offs, count)
        self._keys = K.compact(keys, flags, offs, count)

    def _sort_indices(self):
        self._canonicalize(do_sort=True, do_sum=False)

    def _sum_duplicates(self):
        self._canonicalize(do_sort=False, do_sum=True)

    def _prune(self):
        """Drop stored elements bit-identical to the fill value (reference core.py:1355-1371)."""
        if self.nnz == 0:
            return
        if K.known_eq_bits(self.data, self.fill_value) == 0:     # (the kernel that wrote the values counted its exact zeros: nothing to read)
            return
        if self.nnz >= K.PRUNE_COUNT_FIRST and K.count_eq_bits(self.data, self.fill_value) == 0:
            return
        flags = K.flag_ne_bits(self.data, self.fill_value)
        offs = K.exclusive_scan(flags)
        count = int(offs[-1])
        if count == self.nnz:
            return
        if self.__dict__.get("_coords") is not None:
            self.coords = K.compact(self.coords, flags, offs, count)
        self.data = K.compact(self.data, flags, offs, count)
        if getattr(self, "_keys", None) is not None:
            self._keys = K.compact(self._keys, flags, offs, count)

    # ---- construction / copies ---------------------------------------------------------------
    def _make_shallow_copy_of(self, other):
        self.__dict__["_coords"] = other.__dict__.get("_coords")
        self._coords_dtype = other._index_dtype
        self.data, self.shape = other.data, other.shape
        self.fill_value = other.fill_value
        self._cache = None
        self._keys = getattr(other, "_keys", None)
        from ._dot import drop_derived

        drop_derived(self)   # CSR view / tiled block streams / NaN verdict were built from the old buffers

    def copy(self, deep=True):
        if not deep:
            return _copy.copy(self)
        return COO(self.coords.clone(), self.data.clone(), shape=self.shape, has_duplicates=False, sorted=True,
                   fill_value=self.fill_value)

    def enable_caching(self):
        """Memoise recent transposes/reshapes (reference core.py:317-338)."""
        from collections import OrderedDict

        self._cache = OrderedDict()
        return self

    @classmethod
    def from_numpy(cls, x, fill_value=None, idx_dtype=None, device=None):
        """Dense ndarray / tensor -> COO (reference core.py:341-384): stored elements are those
        NOT bit-identical to the fill value."""
        if isinstance(x, torch.Tensor):
            xt = dev.to_device(x, device if device is not None else (x.device if x.is_cuda else None))
        else:
            x = np.asanyarray(x).view(type=np.ndarray)
            xt = dev.to_device(x, device)
        shape = tuple(int(s) for s in xt.shape)
        npdt = dev.np_dtype(xt.dtype)
        if fill_value is None:
            fill_value = zero_of_dtype(npdt) if shape else npdt.type(xt.item())
        flat = xt.reshape(-1).contiguous()
        it = torch.int64 if idx_dtype is None or np.dtype(idx_dtype).itemsize > 4 else torch.int32
        fused = K.dense_nonfill(flat, fill_value)
        if fused is not None:
            return cls._from_sorted_keys(fused[0], fused[1], shape, fill_value, it)
        flags = K.flag_ne_bits(flat, fill_value)
        offs = K.exclusive_scan(flags)
        count = int(offs[-1])
        n = flat.numel()
        iota = torch.empty(n, dtype=torch.int64, device=flat.device)
        from . import _ffi

        _ffi.call("spamd_iota", n, dev.ptr(iota), dev.stream_ptr(flat.device))
        keys = K.compact(iota, flags, offs, count)
        data = K.compact(flat, flags, offs, count)
        it = torch.int64 if idx_dtype is None or np.dtype(idx_dtype).itemsize > 4 else torch.int32
        return cls._from_sorted_keys(keys, data, shape, fill_value, it)

    @classmethod
    def from_scipy_sparse(cls, x, /, *, fill_value=None, device=None):
        x = x.asformat("coo")
        if not x.has_canonical_format:
            x.eliminate_zeros()
            x.sum_duplicates()
        coords = np.stack([x.row, x.col]).astype(np.int64)
        return cls(coords, x.data, shape=x.shape, has_duplicates=False, sorted=False, fill_value=fill_value,
                   device=device)

    def todense_device(self):
        """Dense tensor in HBM (fill value everywhere, stored values scattered in)."""
        out = torch.full((max(self.size, 1),), self.fill_value.item() if hasattr(self.fill_value, "item") else self.fill_value,
                         dtype=self.data.dtype, device=self.device)
        if self.nnz:
            if self.ndim == 0:
                out[0] = self.data[0]
            else:
                K.scatter_into(out, self.linear_loc(), self.data)
        return out[: self.size].reshape(self.shape) if self.size else out[:0].reshape(self.shape)

    def todense(self):
        """Dense host ndarray (reference core.py:386-422)."""
        return dev.to_numpy(self.todense_device())

    def to_scipy_sparse(self, accept_fv=None):
        import scipy.sparse

        from ._utils import check_fill_value

        check_fill_value(self, accept_fv=accept_fv)
        if self.ndim != 2:
            raise ValueError("Can only convert a 2-dimensional array to a Scipy sparse matrix.")
        c = dev.to_numpy(self.coords)
        return scipy.sparse.coo_matrix((dev.to_numpy(self.data), (c[0], c[1])), shape=self.shape)

    # ---- properties ----------------------------------------------------------------------------
    @property
    def nnz(self):
        c = self.__dict__.get("_coords")
        return int(c.shape[1]) if c is not None else int(self.data.numel())

    @property
    def format(self):
        return "coo"

    @property
    def nbytes(self):
        isz = 8 if self._index_dtype == torch.int64 else 4
        return self.data.numel() * self.data.element_size() + self.nnz * self.ndim * isz

    @property
    def T(self):
        return self.transpose(tuple(range(self.ndim))[::-1])

    @property
    def mT(self):
        if self.ndim < 2:
            raise ValueError("Cannot compute matrix transpose if `ndim < 2`.")
        axis = list(range(self.ndim))
        axis[-1], axis[-2] = axis[-2], axis[-1]
        return self.transpose(axis)

    def __str__(self):
        return (f"<COO: shape={self.shape!s}, dtype={self.dtype!s}, nnz={self.nnz:d}, fill_value={self.fill_value!s}, "
                f"device={self.device}>")

    __repr__ = __str__

    # ---- shape manipulation (keys only; Appendix C.9) ------------------------------------------
    def transpose(self, axes=None):
        """Permute the axes (reference core.py:725-807): permute the keys, stable sort."""
        if axes is None:
            axes = tuple(reversed(range(self.ndim)))
        axes = normalize_axis(tuple(axes), self.ndim)
        if len(set(axes)) != len(axes):
            raise ValueError("repeated axis in transpose")
        if set(axes) != set(range(self.ndim)):
            raise ValueError("axes don't match array")
        axes = tuple(axes)
        if axes == tuple(range(self.ndim)):
            return self
        if self._cache is not None and ("transpose", axes) in self._cache:
            return self._cache[("transpose", axes)]
        shape = tuple(self.shape[ax] for ax in axes)
        keys = K.permute_keys(self.linear_loc(), self.shape, axes)
        keys, perm = K.sort_keys(keys, max(self.size - 1, 1))
        out = COO._from_sorted_keys(keys, K.gather(self.data, perm), shape, self.fill_value, self._index_dtype)
        if self._cache is not None:
            self._cache[("transpose", axes)] = out
            while len(self._cache) > 3:
                self._cache.popitem(last=False)
        return out

    def reshape(self, shape, order="C"):
        """C-order reshape (reference core.py:1034-1111): the linear keys do not change, only
        their split into coordinates; sortedness is preserved."""
        shape = tuple(shape) if isinstance(shape, Iterable) else (shape,)
        if order not in {"C", None}:
            raise NotImplementedError("The `order` parameter is not supported")
        if any(d == -1 for d in shape):
            extra = int(self.size / max(1, prod(d for d in shape if d != -1)))
            shape = tuple(d if d != -1 else extra for d in shape)
        shape = tuple(int(d) for d in shape)
        if self.size != prod(shape):
            raise ValueError(f"cannot reshape array of size {self.size} into shape {shape}")
        if self.shape == shape:
            return self
        if self._cache is not None and ("reshape", shape) in self._cache:
            return self._cache[("reshape", shape)]
        keys = self.linear_loc()
        it = self._index_dtype
        if it == torch.int32 and shape and max(shape) >= 2 ** 31:
            it = torch.int64
        out = COO._from_sorted_keys(keys, self.data, shape, self.fill_value, it)
        if self._cache is not None:
            self._cache[("reshape", shape)] = out
            while len(self._cache) > 3:
                self._cache.popitem(last=False)
        return out

    def flatten(self, order="C"):
        return self.reshape(-1)

    def __getitem__(self, index):
        """Basic indexing (integers, slices, None, Ellipsis); the two forms N-D `matmul` uses — `x[(None,) * k]` and
        `x[i]` — have direct paths, everything else goes through `_indexing.getitem`."""
        if isinstance(index, tuple) and all(i is None for i in index):
            k = len(index)
            if k == 0:
                return self
            lead = torch.zeros((k, self.nnz), dtype=self.coords.dtype, device=self.device)
            out = COO(torch.cat([lead, self.coords]), self.data, shape=(1,) * k + self.shape, has_duplicates=False,
                      sorted=True, fill_value=self.fill_value)
            out._keys = getattr(self, "_keys", None)
            return out
        if isinstance(index, (int, np.integer)) and self.ndim > 1:
            from ._batched import take_leading

            return take_leading(self, int(index))   # contiguous key range: two binary searches, no mask
        from ._indexing import getitem

        return getitem(self, index)

    # ---- format conversion -----------------------------------------------------------------------
    def asformat(self, format, **kwargs):
        from ._utils import convert_format

        format = convert_format(format)
        if format == "coo":
            return self
        if format == "gcxs":
            from ._gcxs import GCXS

            return GCXS.from_coo(self, **kwargs)
        raise NotImplementedError(f"format {format!r} is not available in the hip backend")

    def tocsr(self):
        """`scipy.sparse.csr_array` of a 2-D array (reference core.py:1200-1251); the row compression runs on the
        device, only the three result arrays cross to the host."""
        return self._to_scipy_compressed(0)

    def tocsc(self):
        """`scipy.sparse.csc_array` (reference core.py:1253-1291)."""
        return self._to_scipy_compressed(1)

    def _to_scipy_compressed(self, axis):
        import scipy.sparse

        from ._gcxs import GCXS
        from ._utils import check_zero_fill_value

        check_zero_fill_value(self)
        if self.ndim != 2:
            raise ValueError("This array must be two-dimensional for this conversion to work.")
        g = GCXS.from_coo(self, compressed_axes=(axis,))
        arrays = (g.data.cpu().numpy(), g.indices.cpu().numpy(), g.indptr.cpu().numpy())
        return (scipy.sparse.csr_array if axis == 0 else scipy.sparse.csc_array)(arrays, shape=self.shape)

    @classmethod
    def from_iter(cls, x, shape=None, fill_value=None, dtype=None, device=None):
        """COO from `{(i, j, ...): value}`, an iterable of `((i, j, ...), value)` pairs, or `(data, (rows, cols, ...))`
        (reference `from_iter`, core.py:469-560).  The iterable is a host object; it is packed once and uploaded."""
        from collections.abc import Sized

        if isinstance(x, dict):
            x = list(x.items())
        if not isinstance(x, Sized):
            x = list(x)
        if len(x) != 2 and not all(len(item) == 2 for item in x):
            raise ValueError("Invalid iterable to convert to COO.")
        if not x:
            nd = 0 if shape is None else len(shape)
            coords, data, shape = np.empty((nd, 0), dtype=np.int64), np.empty(0, dtype=dtype), (() if shape is None else shape)
        elif not isinstance(x[0][0], Iterable):
            coords, data = np.stack([np.asarray(c) for c in x[1]], axis=0), np.asarray(x[0], dtype=dtype)
        else:
            coords, data = np.array([item[0] for item in x]).T, np.array([item[1] for item in x], dtype=dtype)
        if not (coords.ndim == 2 and data.ndim == 1 and coords.shape[1] == data.shape[0]
                and np.issubdtype(coords.dtype, np.integer)):
            raise ValueError("Invalid iterable to convert to COO.")
        return cls(coords, data, shape=shape, fill_value=fill_value, device=device)

    def broadcast_to(self, shape):
        from ._broadcast import broadcast_to

        return broadcast_to(self, shape)

    def nonzero(self):
        """Coordinates of the stored elements, one device array per dimension (reference core.py `nonzero`)."""
        from ._utils import check_zero_fill_value

        check_zero_fill_value(self)
        if self.ndim == 0:
            raise ValueError("`nonzero` is undefined for `self.ndim == 0`.")
        return tuple(self.coords)

    def swapaxes(self, axis1, axis2):
        axes = list(range(self.ndim))
        axes[axis1], axes[axis2] = axes[axis2], axes[axis1]
        return self.transpose(axes)

    def squeeze(self, axis=None):
        from ._api import squeeze

        return squeeze(self, axis=axis)

    # pickling: the state is host arrays (a pickle must not depend on the GPU it was written from); loading puts
    # the array on the current default device (reference core.py:293-298 keeps the same 4-tuple)
    def __getstate__(self):
        return (self.coords.cpu().numpy(), self.data.cpu().numpy(), self.shape, self.fill_value)

    def __setstate__(self, state):
        coords, data, shape, fill_value = state
        d = dev.default_device()
        self.data = dev.to_device(data, d)
        self.__dict__["_coords"] = torch.from_numpy(np.ascontiguousarray(coords)).to(d)
        self._coords_dtype = self.__dict__["_coords"].dtype
        self.shape, self.fill_value = tuple(shape), fill_value
        self._keys = None
        self._cache = None

    def maybe_densify(self, max_size=1000, min_density=0.25):
        if self.size <= max_size or self.density >= min_density:
            return self.todense()
        raise ValueError("Operation would require converting large sparse array to dense")

    def dot(self, other):
        from ._dot import dot

        return dot(self, other)

    def __matmul__(self, other):
        return _dot_module().matmul(self, other)

    def __rmatmul__(self, other):
        return _dot_module().matmul(other, self)


def as_coo(x, shape=None, fill_value=None, idx_dtype=None, device=None):
    """Convert to `COO` (reference `as_coo`, _coo/core.py)."""
    from ._gcxs import GCXS

    if isinstance(x, COO):
        return x
    if isinstance(x, GCXS):
        return x.tocoo()
    if isinstance(x, SparseArray):
        return x.asformat("coo")
    if isinstance(x, (np.ndarray, torch.Tensor)) or np.isscalar(x):
        if shape is not None:
            raise ValueError("Cannot get `shape` or `fill_value` from a dense array")
        return COO.from_numpy(np.asarray(x) if np.isscalar(x) else x, fill_value=fill_value, idx_dtype=idx_dtype,
                              device=device)
    if _is_scipy_sparse(x):
        return COO.from_scipy_sparse(x, device=device)
    raise NotImplementedError(f"Format not supported for conversion. Supplied type is {type(x)}")


_DOT = None


def _dot_module():
    """`._dot`, imported at first use (it imports this module) and kept: a function-local `from ._dot import matmul` costs
    ~1 us per `a @ b`, which is visible in products of a few hundred stored elements"""
    global _DOT
    if _DOT is None:
        from . import _dot

        _DOT = _dot
    return _DOT
