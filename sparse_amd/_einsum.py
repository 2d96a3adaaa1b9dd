"""`einsum` over sparse operands (reference `einsum`, _common.py:1163-1476).

Semantics follow the reference: every term is first reduced to the labels it shares with another term or
with the output (diagonals of repeated labels are selected, labels that appear nowhere else are summed
out), then the terms are combined and a final single-term pass produces the requested label order.

How the terms are combined is chosen per call:
  * two terms whose shared labels are all summed out (no batch label) go through `tensordot` — the hot path
    (A1/A3/A4 kernels), no outer product is formed;
  * anything else takes the reference's route: reshape every term to the aligned label set with size-1
    axes, broadcast-multiply, reduce.
All arithmetic runs on the device (COO canonicalisation sums duplicates left to right like the reference's
`COO(..., has_duplicates=True)`).
"""
import string
from functools import reduce as _fold
from operator import mul as _mul

import numpy as np
import torch

from . import _kernels as K
from ._sparse_array import SparseArray
from ._utils import check_zero_fill_value

_LABELS = string.ascii_uppercase + string.ascii_lowercase


def _subscripts_from_lists(operands):
    """einsum(op0, sublist0, op1, sublist1, ..., [sublistout]) -> ('ab,bc->ac', [op0, op1, ...])"""
    arrays, parts = [], []
    rest = list(operands)
    while len(rest) >= 2:
        arrays.append(rest.pop(0))
        parts.append(rest.pop(0))
    def text(sub):
        if not isinstance(sub, (list, tuple)):
            raise TypeError("For this input type lists must contain either int or Ellipsis")
        out = ""
        for s in sub:
            if s is Ellipsis:
                out += "..."
                continue
            try:
                out += _LABELS[int(np.asarray(s).astype(np.int64)) if not isinstance(s, int) else s]
            except (TypeError, ValueError, IndexError) as e:
                raise TypeError("For this input type lists must contain either int or Ellipsis") from e
        return out

    subs = ",".join(text(p) for p in parts)
    if rest:
        subs += "->" + text(rest[0])
    return subs, arrays


def parse_einsum(operands):
    """Normalise the call to (terms, output, arrays): one label string per operand with every ellipsis
    replaced by explicit labels, and the explicit output labels (classical implicit rule applied)."""
    if len(operands) == 0:
        raise ValueError("No input operands")
    if isinstance(operands[0], str):
        subs, arrays = operands[0].replace(" ", ""), list(operands[1:])
        for ch in subs:
            if ch not in ".,->" and not ch.isalpha():
                raise ValueError(f"Character {ch} is not a valid symbol.")
    else:
        subs, arrays = _subscripts_from_lists(operands)
    if subs.count("->") > 1 or ("-" in subs) != (">" in subs):
        raise ValueError("Subscripts can only contain one '->'.")
    lhs, arrow, out = subs.partition("->")
    terms = lhs.split(",")
    if len(terms) != len(arrays):
        raise ValueError("Number of einsum subscripts must be equal to the number of operands.")
    used = set(subs) - set(".,->")
    spare = [c for c in _LABELS if c not in used]
    longest = 0
    expanded = []
    for term, arr in zip(terms, arrays):
        nd = len(getattr(arr, "shape", ()))
        if "." in term:
            if term.count(".") != 3 or "..." not in term:
                raise ValueError("Invalid Ellipses.")
            n_ell = nd - (len(term) - 3)
            if n_ell < 0:
                raise ValueError("Ellipses lengths do not match.")
            longest = max(longest, n_ell)
            # ellipsis axes align to the right (broadcasting): take the LAST n_ell spare labels
            term = term.replace("...", "".join(spare[len(spare) - n_ell:]) if n_ell else "")
        elif len(term) != nd:
            raise ValueError(f"einstein sum subscripts string contains too many subscripts for operand"
                             if len(term) > nd else
                             "operand has more dimensions than subscripts given in einstein sum, but no '...' ellipsis "
                             "provided to broadcast the extra dimensions.")
        expanded.append(term)
    ell = "".join(spare[len(spare) - longest:]) if longest else ""
    if arrow:
        if "." in out:
            if out.count(".") != 3 or "..." not in out:
                raise ValueError("Invalid Ellipses.")
            out = out.replace("...", ell)
        for ch in out:
            if out.count(ch) != 1:
                raise ValueError(f"Output character {ch} appeared more than once in the output.")
            if not any(ch in t for t in expanded):
                raise ValueError(f"Output character {ch} did not appear in the input")
    else:
        joined = "".join(expanded)
        out = ell + "".join(sorted(c for c in set(joined) if joined.count(c) == 1 and c not in ell))
    return expanded, out, arrays


def _single(term, out, x):
    """One operand: diagonals of repeated labels, summation of dropped labels, transposition."""
    from ._coo import COO, as_coo
    from ._umath import binary_arrays

    if term == out:
        return x.sum() if not out else x
    if not isinstance(x, SparseArray):
        return np.einsum(f"{term}->{out}", np.asarray(x.cpu()) if isinstance(x, torch.Tensor) else x)
    back = getattr(x, "from_coo", None) if not isinstance(x, COO) else None
    x = as_coo(x)
    where = {}
    for axis, label in enumerate(term):
        where.setdefault(label, []).append(axis)
    coords, data = x.coords, x.data
    keep = None
    for label, axes in where.items():
        if len(axes) > 1:
            if len({x.shape[a] for a in axes}) > 1:
                raise ValueError("Repeated indices must have the same dimension.")
            for a in axes[1:]:
                same = binary_arrays("equal", coords[axes[0]].contiguous(), coords[a].contiguous(), out_bool_as=torch.uint8)
                keep = same if keep is None else binary_arrays("logical_and", keep, same, out_bool_as=torch.uint8)
    if keep is not None and x.nnz:
        flags = K.flag_ne_bits(keep, 0)
        offs = K.exclusive_scan(flags)
        count = int(offs[-1])
        coords = K.compact(coords, flags, offs, count)
        data = K.compact(data, flags, offs, count)
    order = [term.index(label) for label in out]
    if not out:   # full contraction: every kept element lands on the single position of a 0-d result
        n = int(data.numel())
        line = COO(torch.zeros((1, n), dtype=torch.int64, device=x.device), data, shape=(1,), has_duplicates=True)
        return line.sum()
    picked = torch.stack([coords[a] for a in order]) if order else coords[:0]   # row selection: memory plumbing
    res = COO(picked, data, shape=tuple(x.shape[a] for a in order), has_duplicates=True)
    return back(res) if back is not None else res


def _pair_by_tensordot(terms, out, arrays):
    """Two terms, no repeated label inside a term, every shared label summed out: a tensordot plus a
    transposition.  Returns None when the pattern does not apply."""
    from ._coo import COO
    from ._dot import tensordot
    from ._gcxs import GCXS

    ta, tb = terms
    if len(set(ta)) != len(ta) or len(set(tb)) != len(tb):
        return None
    shared = [c for c in ta if c in tb]
    if not shared or any(c in out for c in shared):
        return None
    free = [c for c in ta if c not in shared] + [c for c in tb if c not in shared]
    if sorted(free) != sorted(out):
        return None   # some free label is summed out as well: leave it to the general route
    a, b = arrays
    for c in shared:
        if a.shape[ta.index(c)] != b.shape[tb.index(c)]:
            raise ValueError(f"Inconsistent shape for index '{c}'.")
    if isinstance(a, SparseArray) and isinstance(b, SparseArray) and (len(shared) == len(ta) or len(shared) == len(tb)):
        # a SPARSE operand that is contracted away entirely ("ij,ij->", "ijk,jk->i"): as a matrix it is one row (or one
        # column) over the product of the contracted extents - 10^9 row pointers for two 10^5 x 10^4 operands, 929 ms - while
        # the general route multiplies the aligned operands and sums (0.7 ms): late round 6, tools/r06/einsum_sweep.py
        return None
    res = tensordot(a, b, axes=([ta.index(c) for c in shared], [tb.index(c) for c in shared]))
    # result format as the reference's broadcast-multiply route gives it: sparse, GCXS only if every sparse
    # operand is GCXS
    sparse_ops = [x for x in arrays if isinstance(x, SparseArray)]
    if not isinstance(res, SparseArray):
        res = COO.from_numpy(res if isinstance(res, torch.Tensor) else np.asarray(res), device=sparse_ops[0].device)
    if out:
        perm = [free.index(c) for c in out]
        if perm != list(range(len(perm))):
            res = res.transpose(perm)
    want = "gcxs" if all(isinstance(x, GCXS) for x in sparse_ops) else "coo"
    return res if res.format == want or res.ndim == 0 else res.asformat(want)


def einsum(*operands, **kwargs):
    """`numpy.einsum` for sparse (and mixed sparse/dense) operands (reference `einsum`, _common.py:1400-1476)."""
    terms, out, arrays = parse_einsum(operands)
    check_zero_fill_value(*[a for a in arrays if isinstance(a, SparseArray)])
    dtype = kwargs.pop("dtype", None)
    kwargs.pop("optimize", None)
    if kwargs:
        raise TypeError(f"einsum() got unexpected keyword arguments {sorted(kwargs)}")
    if dtype is not None:
        arrays = [a.astype(dtype) for a in arrays]
    if len(arrays) == 1:
        return _single(terms[0], out, arrays[0])
    if not any(isinstance(a, SparseArray) for a in arrays):
        raise ValueError(f"None of the args is sparse: {arrays}")
    sizes, seen_in = {}, {}
    for t, (term, arr) in enumerate(zip(terms, arrays)):
        for label, extent in zip(term, arr.shape):
            if sizes.setdefault(label, int(extent)) != int(extent):
                raise ValueError(f"Inconsistent shape for index '{label}'.")
            seen_in.setdefault(label, set()).add(t)
    for label in out:
        seen_in[label].add(-1)
    if len(arrays) == 2 and all(isinstance(a, (SparseArray, np.ndarray, torch.Tensor)) for a in arrays) \
            and any(isinstance(a, SparseArray) for a in arrays):
        fast = _pair_by_tensordot(terms, out, arrays)
        if fast is not None:
            return fast
    aligned = "".join(label for label, users in seen_in.items() if len(users) > 1)
    prepared = []
    for term, arr in zip(terms, arrays):
        mine = "".join(label for label in aligned if label in term)
        if mine != term:
            arr = _single(term, mine, arr)
        shape = tuple(arr.shape[mine.index(label)] if label in mine else 1 for label in aligned)
        prepared.append(arr.reshape(shape) if tuple(arr.shape) != shape else arr)
    return _single(aligned, out, _fold(_mul, prepared))
