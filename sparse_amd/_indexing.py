"""Basic indexing of COO / GCXS arrays: integers, slices (any step), `None`, `Ellipsis`
(reference `_coo/indexing.py:12-133`, `_compressed/indexing.py:14-176`).

The reference narrows the sorted coordinate list dimension by dimension with binary searches.  Here every
indexed dimension contributes one device mask over the stored elements (equality for an integer, a range +
stride test for a slice); the masks are AND-ed, the survivors compacted, and the slice coordinates rebased with
integer arithmetic — all through the elementwise/compaction kernels of libsparse_amd.so.  Advanced (array)
indices are not on this path and raise NotImplementedError."""
import numpy as np
import torch

from . import _kernels as K


def _normalise(index, ndim):
    if not isinstance(index, tuple):
        index = (index,)
    if sum(1 for i in index if i is Ellipsis) > 1:
        raise IndexError("an index can only have a single ellipsis ('...')")
    for i in index:
        if not (i is None or i is Ellipsis or isinstance(i, (int, np.integer, slice))):
            if isinstance(i, (list, np.ndarray, torch.Tensor)) or hasattr(i, "coords"):
                raise NotImplementedError("advanced (array) indexing is outside the hip backend's hot path (SURVEY.md §8f N2)")
            raise IndexError("only integers, slices (`:`), ellipsis (`...`), numpy.newaxis (`None`) are valid indices here")
    real = sum(1 for i in index if i is not None and i is not Ellipsis)
    if real > ndim:
        raise IndexError(f"too many indices for array: array is {ndim}-dimensional, but {real} were indexed")
    rest = (slice(None),) * (ndim - real)
    if Ellipsis in index:
        at = index.index(Ellipsis)
        return index[:at] + rest + index[at + 1:]
    return index + rest


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
    dim = 0
    for it in index:
        if it is None:
            plan.append(None)
            shape.append(1)
            continue
        n = c.shape[dim]
        row = coords[dim].contiguous() if c.nnz else None
        if isinstance(it, slice):
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
