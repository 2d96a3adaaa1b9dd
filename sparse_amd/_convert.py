"""Device-side format conversion.

The reference routes every COO<->GCXS conversion, `change_compressed_axes`, N-D `reshape` and
`transpose` through one per-element index re-linearisation + a stable argsort
(`_compressed/convert.py:210-339`, `_compressed/compressed.py:25-77,388-460`).  Here the same
thing is expressed on 64-bit linear keys: permute keys (csrc/prims.hip) -> stable radix sort
-> split into (indptr, indices) or coordinates.
"""
import numpy as np
import torch

from . import _kernels as K
from ._utils import can_store, check_compressed_axes, normalize_axis, prod


def _axis_order(ndim, compressed_axes):
    order = list(compressed_axes)
    order.extend(a for a in range(ndim) if a not in compressed_axes)
    return order


def _inverse(perm):
    inv = [0] * len(perm)
    for d, s in enumerate(perm):
        inv[s] = d
    return inv


def _pick_index_dtype(base, bound):
    if base == torch.int32 and bound < 2 ** 31:
        return torch.int32
    return torch.int64


def coo_to_gcxs_arrays(x, compressed_axes=None, idx_dtype=None):
    """`_from_coo` (reference _compressed/compressed.py:25-77): returns
    ((data, indices, indptr), shape, compressed_axes, fill_value)."""
    if x.ndim == 0:
        if compressed_axes is not None:
            raise ValueError("no axes to compress for 0d array")
        return ((x.data, x.coords, []), x.shape, None, x.fill_value)
    if x.ndim == 1:
        if compressed_axes is not None:
            raise ValueError("no axes to compress for 1d array")
        return ((x.data, x.coords[0], ()), x.shape, None, x.fill_value)
    compressed_axes = normalize_axis(compressed_axes, x.ndim)
    if compressed_axes is None:
        compressed_axes = (int(np.argmin(x.shape)),)  # best compression ratio (reference :37-39)
    check_compressed_axes(x.shape, compressed_axes)
    order = _axis_order(x.ndim, compressed_axes)
    rshape = tuple(x.shape[i] for i in order)
    R = prod(rshape[: len(compressed_axes)])
    C = prod(rshape[len(compressed_axes):])
    bound = max(R, C, x.nnz)
    if idx_dtype and not can_store(idx_dtype, bound):
        raise ValueError(f"cannot store array with the compressed shape {(R, C)} and nnz {x.nnz} with dtype {idx_dtype}.")
    base = x.coords.dtype if not idx_dtype else (torch.int32 if np.dtype(idx_dtype).itemsize <= 4 else torch.int64)
    it = _pick_index_dtype(base, bound)
    if x.ndim == 2 and tuple(compressed_axes) == (1,) and x.data.element_size() == 4 and x.data.dtype != torch.bool \
            and max(x.shape) < 2 ** 31 and x.nnz < 2 ** 31 and x.coords.dtype in (torch.int32, torch.int64):
        # canonical 2-D COO is CSR order already: row pointers + the one-call CSR -> CSC swap (a stable sort on the
        # column alone) instead of re-linearising and sorting the full (column, row) keys
        ct = x.coords.dtype
        rows_ptr = K.rows_to_indptr(x.coords[0], int(x.shape[0])).to(ct)
        data, indices, indptr = K.csx_swap_2d(x.data, x.coords[1].contiguous(), rows_ptr, int(x.shape[0]), int(x.shape[1]))
        return ((data, indices.to(it), indptr.to(it)), x.shape, tuple(compressed_axes), x.fill_value)
    keys = x.linear_loc()
    data = x.data
    if order != list(range(x.ndim)):
        keys = K.permute_keys(keys, x.shape, order)
        if data.element_size() in (4, 8) and data.dtype != torch.bool:
            keys, data = K.sort_key_value(keys, data, max(x.size - 1, 1))  # the values ride along as the payload
        else:
            keys, perm = K.sort_keys(keys, max(x.size - 1, 1))
            data = K.gather(data, perm)
    indptr, indices = K.keys_to_csr(keys, R, C, it)
    return ((data, indices, indptr), x.shape, tuple(compressed_axes), x.fill_value)


def gcxs_natural_keys(x):
    """C-order linear keys (natural axis order) of every stored element of a GCXS, in storage
    order — `uncompress_dimension` + un-reordering (reference convert.py:82-87, compressed.py:448-460)."""
    if x.ndim == 1:
        return x.indices.to(torch.int64)
    R, C = x._compressed_shape
    keys = K.csr_to_keys(x.indptr, x.indices, R, C)
    order = x._axis_order
    if order != list(range(x.ndim)):
        keys = K.permute_keys(keys, x._reordered_shape, _inverse(order))
    return keys


def gcxs_to_coo(x):
    """`GCXS.tocoo` (reference compressed.py:425-460)."""
    from ._coo import COO

    if x.ndim == 0:
        return COO(torch.zeros((0, x.nnz), dtype=torch.int64, device=x.device), x.data, shape=x.shape,
                   fill_value=x.fill_value)
    if x.ndim == 1:
        return COO(x.indices[None, :], x.data, shape=x.shape, fill_value=x.fill_value)
    keys = gcxs_natural_keys(x)
    unsorted, _ = K.keys_check(keys)
    data = x.data
    if unsorted:
        keys, perm = K.sort_keys(keys, max(x.size - 1, 1))
        data = K.gather(data, perm)
    it = x.indices.dtype if max(x.shape) < 2 ** 31 or x.indices.dtype == torch.int64 else torch.int64
    # (sorted, duplicate-free keys: the coordinates are split off when somebody asks for them)
    return COO._from_sorted_keys(keys, data, x.shape, x.fill_value, it)


def gcxs_relayout(x, shape, axes, compressed_axes, transpose=False, reshape=False):
    """The one routine behind `change_compressed_axes`, N-D `transpose` and `reshape`
    (`convert._transpose`, reference _compressed/convert.py:210-273): returns the new
    (data, indices, indptr)."""
    shape = tuple(int(s) for s in shape)
    if (x.ndim == 2 and not transpose and not reshape and len(x.compressed_axes) == 1 and len(compressed_axes) == 1
            and tuple(compressed_axes) != tuple(x.compressed_axes)):
        c = x.compressed_axes[0]
        return K.csx_swap_2d(x.data, x.indices, x.indptr, shape[c], shape[1 - c])
    keys = gcxs_natural_keys(x)
    nat_shape = x.shape
    if transpose:
        keys = K.permute_keys(keys, x.shape, axes)
        nat_shape = tuple(x.shape[a] for a in axes)
    if reshape:
        nat_shape = shape  # C-order reshape: linear keys are unchanged
    assert tuple(nat_shape) == shape
    order = _axis_order(len(shape), compressed_axes)
    rshape = tuple(shape[i] for i in order)
    R = prod(rshape[: len(compressed_axes)])
    C = prod(rshape[len(compressed_axes):])
    keys = K.permute_keys(keys, shape, order)
    if x.data.element_size() in (4, 8) and x.data.dtype != torch.bool:
        keys, data = K.sort_key_value(keys, x.data, max(prod(shape) - 1, 1))  # the values ride along as the payload
    else:
        keys, perm = K.sort_keys(keys, max(prod(shape) - 1, 1))
        data = K.gather(x.data, perm)
    it = _pick_index_dtype(x.indices.dtype, max(R, C, x.nnz))
    indptr, indices = K.keys_to_csr(keys, R, C, it)
    return (data, indices, indptr)


def gcxs_todense(x):
    """Dense device tensor of a GCXS (reference compressed.py:462-482)."""
    fv = x.fill_value.item() if hasattr(x.fill_value, "item") else x.fill_value
    out = torch.full((max(x.size, 1),), fv, dtype=x.data.dtype, device=x.device)
    if x.nnz:
        if x.ndim == 0:
            out[0] = x.data[0]
        else:
            K.scatter_into(out, gcxs_natural_keys(x), x.data)
    return out[: x.size].reshape(x.shape)


def gcxs_prune(x):
    """Remove stored fill values, rebuilding indptr (reference compressed.py:816-848)."""
    if x.nnz == 0:
        return
    if K.known_eq_bits(x.data, x.fill_value) == 0:     # (the kernel that wrote the values counted its exact zeros: nothing to read)
        return
    if x.nnz >= K.PRUNE_COUNT_FIRST and K.count_eq_bits(x.data, x.fill_value) == 0:
        return
    flags = K.flag_ne_bits(x.data, x.fill_value)
    offs = K.exclusive_scan(flags)
    count = int(offs[-1])
    if count == x.nnz:
        return
    x.data = K.compact(x.data, flags, offs, count)
    x.indices = K.compact(x.indices, flags, offs, count)
    if x.indptr.numel():
        x.indptr = K.gather(offs, x.indptr.to(torch.int64)).to(x.indptr.dtype)
