"""Indexing of COO / GCXS arrays: integers, slices (any step), `None`, `Ellipsis` and ONE 1-D integer (or boolean) array
(reference `_coo/indexing.py:12-133`, `_compressed/indexing.py:14-176`).

The reference narrows the sorted coordinate list dimension by dimension with binary searches.  Here every
indexed dimension contributes one device mask over the stored elements (equality for an integer, a range +
stride test for a slice); the masks are AND-ed, the survivors compacted, and the slice coordinates rebased with
integer arithmetic — all through the elementwise/compaction kernels of libsparse_amd.so.  An integer array index
(`take`) is a join of the survivors' coordinates with the index array: the array is sorted once, two binary searches per
stored element give the run of positions that ask for its coordinate, and the element is replicated that often
(scan + expansion).  Several array indices at once are not on this path and raise NotImplementedError."""
import numpy as np
import torch

from . import _kernels as K


def _normalise(index, ndim):
    if not isinstance(index, tuple):
        index = (index,)
    if sum(1 for i in index if i is Ellipsis) > 1:
        raise IndexError("an index can only have a single ellipsis ('...')")
    index = tuple(_as_index_array(i) if isinstance(i, (list, np.ndarray, torch.Tensor)) else i for i in index)
    if sum(1 for i in index if isinstance(i, np.ndarray)) > 1:
        raise NotImplementedError("several array indices at once are outside the hip backend's path (SURVEY.md §8f N2)")
    for i in index:
        if not (i is None or i is Ellipsis or isinstance(i, (int, np.integer, slice, np.ndarray))):
            if hasattr(i, "coords"):
                raise NotImplementedError("sparse arrays as indices are outside the hip backend's path (SURVEY.md §8f N2)")
            raise IndexError("only integers, slices (`:`), ellipsis (`...`), numpy.newaxis (`None`) and integer or boolean "
                             "arrays are valid indices")
    real = sum(1 for i in index if i is not None and i is not Ellipsis)
    if real > ndim:
        raise IndexError(f"too many indices for array: array is {ndim}-dimensional, but {real} were indexed")
    rest = (slice(None),) * (ndim - real)
    at = next((k for k, i in enumerate(index) if i is Ellipsis), None)     # (identity: `in` would compare array indices)
    if at is not None:
        return index[:at] + rest + index[at + 1:]
    return index + rest


def _as_index_array(i):
    """list / ndarray / tensor index -> 1-D host integer array (a boolean mask selects its True positions)"""
    if isinstance(i, torch.Tensor):
        i = i.cpu().numpy()
    a = np.asarray(i)
    if a.ndim != 1:
        raise NotImplementedError("only 1-D array indices are on the hip backend's path (SURVEY.md §8f N2)")
    if a.dtype == bool:
        return a
    if a.size and not np.issubdtype(a.dtype, np.integer):
        raise IndexError("arrays used as indices must be of integer (or boolean) type")
    return a.astype(np.int64)


def _join_with_index_array(coords, data, dim, idx, n, dev):
    """Every (stored element, position j) with coords[dim] == idx[j]: (coords with row `dim` replaced by j, data)."""
    from ._umath import binary_arrays

    n_kept, L = int(data.numel()), int(idx.size)
    if n_kept == 0 or L == 0:
        return coords[:, :0], data[:0]
    want = torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
    sorted_idx, where = K.sort_keys(want, max(n - 1, 1))
    c = K.convert(coords[dim].contiguous(), torch.int64)
    one = torch.tensor([1], dtype=torch.int64, device=dev)
    lo = torch.empty(n_kept, dtype=torch.int64, device=dev)
    hi = torch.empty(n_kept, dtype=torch.int64, device=dev)
    scratch = torch.empty(n_kept, dtype=torch.int64, device=dev)
    from . import _ffi
    from ._device import ptr, stream_ptr

    s = stream_ptr(dev)
    _ffi.call("spamd_lower_bound_match", n_kept, ptr(c), L, ptr(sorted_idx), ptr(lo), ptr(scratch), s)
    c1 = binary_arrays("add", c, one, b_scalar=True)
    _ffi.call("spamd_lower_bound_match", n_kept, ptr(c1), L, ptr(sorted_idx), ptr(hi), ptr(scratch), s)
    copies = torch.cat([binary_arrays("subtract", hi, lo), one])          # (+ the scan's ignored last slot)
    offs = K.exclusive_scan(copies)
    total = int(offs[-1])
    if total == 0:
        return coords[:, :0], data[:0]
    # owner of every output slot: the CSR expansion with `offs` as row pointers (column part zero)
    owner = K.csr_to_keys(offs, torch.zeros(total, dtype=torch.int64, device=dev), n_kept, 1)
    iota = torch.empty(total, dtype=torch.int64, device=dev)
    _ffi.call("spamd_iota", total, ptr(iota), s)
    nth = binary_arrays("subtract", iota, K.gather(offs, owner))
    j = K.gather(where, binary_arrays("add", K.gather(lo, owner), nth))
    out = K.gather(coords.contiguous(), owner)
    out[dim] = j.to(out.dtype)
    return out, K.gather(data.contiguous(), owner)


def getitem(x, index):
    from ._coo import COO
    from ._gcxs import GCXS
    from ._umath import binary_arrays

    was_gcxs = isinstance(x, GCXS)
    c = x.tocoo() if was_gcxs else x
    index = _normalise(index, c.ndim)
    dev = c.device
    coords = c.coords

    def scalar(v, like):
        return torch.tensor([v], dtype=like.dtype, device=dev)

    def both(a, b):
        return b if a is None else binary_arrays("logical_and", a, b, out_bool_as=torch.uint8)

    keep, plan, shape, ordered = None, [], [], True
    adv = None
    dim = 0
    for it in index:
        if it is None:
            plan.append(None)
            shape.append(1)
            continue
        n = c.shape[dim]
        row = coords[dim].contiguous() if c.nnz else None
        if isinstance(it, np.ndarray):
            if it.dtype == bool:
                if len(it) != n:
                    raise IndexError(f"boolean index did not match indexed array along axis {dim}; size of axis is {n} but size of "
                                     f"corresponding boolean axis is {len(it)}")
                it = np.flatnonzero(it).astype(np.int64)
            if it.size and (it.min() < -n or it.max() >= n):
                bad = int(it[(it < -n) | (it >= n)][0])
                raise IndexError(f"index {bad} is out of bounds for axis {dim} with size {n}")
            adv = (dim, np.where(it < 0, it + n, it), n)
            plan.append((dim, 0, 1))
            shape.append(int(it.size))
            ordered = False
        elif isinstance(it, slice):
            start, stop, step = it.indices(n)
            shape.append(len(range(start, stop, step)))
            if (start, stop, step) == (0, n, 1):
                plan.append((dim, 0, 1))
            else:
                if row is not None:
                    lo, hi = (start, stop) if step > 0 else (stop + 1, start + 1)   # occupied range [lo, hi)
                    m = both(binary_arrays("greater_equal", row, scalar(lo, row), b_scalar=True, out_bool_as=torch.uint8),
                             binary_arrays("less", row, scalar(hi, row), b_scalar=True, out_bool_as=torch.uint8))
                    if abs(step) != 1:   # (row - start) must be a multiple of the step: compare q * step with the offset
                        off = binary_arrays("subtract", row, scalar(start, row), b_scalar=True)
                        q = binary_arrays("divide", off, scalar(step, row), b_scalar=True)
                        m = both(m, binary_arrays("equal", binary_arrays("multiply", q, scalar(step, row), b_scalar=True), off,
                                                  out_bool_as=torch.uint8))
                    keep = both(keep, m)
                plan.append((dim, start, step))
                ordered = ordered and step > 0
        else:
            i = int(it)
            if i < 0:
                i += n
            if not 0 <= i < n:
                raise IndexError(f"index {int(it)} is out of bounds for axis {dim} with size {n}")
            if row is not None:
                keep = both(keep, binary_arrays("equal", row, scalar(i, row), b_scalar=True, out_bool_as=torch.uint8))
        dim += 1

    data = c.data
    if keep is not None and c.nnz:
        flags = K.flag_ne_bits(keep, 0)
        offs = K.exclusive_scan(flags)
        count = int(offs[-1])
        coords = K.compact(coords, flags, offs, count)
        data = K.compact(data, flags, offs, count)
    if adv is not None:
        coords, data = _join_with_index_array(coords, data, adv[0], adv[1], adv[2], dev)
    n_kept = int(data.numel())
    rows = []
    for p in plan:
        if p is None:
            rows.append(torch.zeros(n_kept, dtype=coords.dtype, device=dev))
            continue
        d, start, step = p
        r = coords[d]
        if n_kept and (start, step) != (0, 1):
            r = binary_arrays("subtract", r.contiguous(), scalar(start, r), b_scalar=True)
            if step != 1:   # exact division: the mask kept multiples of the step only
                r = binary_arrays("divide", r, scalar(step, r), b_scalar=True)
        rows.append(r)
    if not shape:   # every axis indexed by an integer: the element itself (value or fill value)
        return (data.cpu().numpy()[0] if n_kept else np.asarray(c.fill_value, dtype=c.dtype)[()])
    new_coords = torch.stack(rows) if rows else coords[:0]
    out = COO(new_coords, data, shape=tuple(shape), has_duplicates=False, sorted=ordered, fill_value=c.fill_value)
    return out.asformat("gcxs") if was_gcxs else out
