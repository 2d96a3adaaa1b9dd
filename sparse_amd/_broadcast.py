"""Sparse (x) sparse broadcasting (reference _umath.py:95-389 `broadcast_to`
and the reduced-coordinate matching of `_match_coo`).

Round 5: targets whose axes are (broadcast)(own)(broadcast)(own)(broadcast) groups - any of them empty: a leading, a middle and
a trailing group of broadcast axes - are materialised by ONE kernel that writes the replicas already sorted
(csrc/broadcast.hip, `_broadcast_groups`); the general construction below (outer sum of keys + a sort) takes the rest.

`broadcast_to` materialises the broadcast array on the device: with base[e] = key of entry e
under the target strides (size-1 axes contribute 0) and off[r] = key offset of replica r along
the broadcast axes, the result's keys are the outer sum base[e] + off[r] (then one stable sort).
Elementwise ops on operands of different shapes broadcast both sides and take the same-shape
union path; results equal the reference's mask-matching (which only avoids the materialisation)."""
import numpy as np
import torch

from . import _ffi
from . import _kernels as K
from ._device import ptr, stream_ptr


def broadcast_shapes(*shapes):
    try:
        return tuple(int(s) for s in np.broadcast_shapes(*shapes))
    except ValueError:
        raise ValueError(f"operands could not be broadcast together with shapes {' '.join(str(s) for s in shapes)}") from None


BROADCAST_FUSED = True
BROADCAST_STATS = {}


def _broadcast_groups(xs, shape):
    """(b0, k1, b1, k2, b2) when the target's axes form at most the groups [B0][K1][B1][K2][B2], else None.  Axes of size 1 in
    both shapes belong to whatever group they stand in."""
    kinds = []      # [kind, size] of maximal groups; kind True = broadcast axis of x
    for a, t in zip(xs, shape):
        if a == 1 and t == 1:
            continue
        kind = a == 1
        if kinds and kinds[-1][0] == kind:
            kinds[-1][1] *= t
        else:
            kinds.append([kind, t])
    sizes = {"b2": 1, "k2": 1, "b1": 1, "k1": 1, "b0": 1}
    for name, want in (("b2", True), ("k2", False), ("b1", True), ("k1", False), ("b0", True)):
        if kinds and kinds[-1][0] == want:
            sizes[name] = kinds.pop()[1]
    if kinds:
        return None
    return sizes["b0"], sizes["k1"], sizes["b1"], sizes["k2"], sizes["b2"]


def broadcast_to(x, shape):
    """COO broadcast to `shape` (reference `broadcast_to`, _umath.py:344-389)."""
    from ._coo import COO
    from ._umath import binary_arrays

    shape = tuple(int(s) for s in shape)
    if x.shape == shape:
        return x
    if broadcast_shapes(x.shape, shape) != shape:
        raise ValueError(f"The shapes {x.shape} and {shape} are not broadcastable.")
    nd = len(shape)
    xs = (1,) * (nd - x.ndim) + tuple(x.shape)
    bdims = [d for d in range(nd) if xs[d] == 1 and shape[d] != 1]
    rep = 1
    for d in bdims:
        rep *= shape[d]
    dev = x.device
    strides = K.c_strides(shape)
    nnz = x.nnz
    if nnz == 0 or rep == 0 or any(s == 0 for s in shape):
        return COO(torch.zeros((nd, 0), dtype=x.coords.dtype, device=dev), x.data[:0], shape=shape,
                   has_duplicates=False, sorted=True, fill_value=x.fill_value)
    groups = _broadcast_groups(xs, shape) if BROADCAST_FUSED and x.data.element_size() in (1, 2, 4, 8) else None
    if groups is not None and nnz * rep < 2 ** 31 * 256:
        b0, k1, b1, k2, b2 = groups
        n_out = nnz * rep
        keys = torch.empty(n_out, dtype=torch.int64, device=dev)
        data = torch.empty(n_out, dtype=x.data.dtype, device=dev)
        _ffi.call("spamd_coo_broadcast", x.data.element_size(), nnz, ptr(x.linear_loc().contiguous()), ptr(x.data.contiguous()),
                  b0, k1, b1, k2, b2, ptr(keys), ptr(data), stream_ptr(dev))
        BROADCAST_STATS["fused"] = BROADCAST_STATS.get("fused", 0) + 1
        it = x.coords.dtype if max(shape) < 2 ** 31 or x.coords.dtype == torch.int64 else torch.int64
        return COO._from_sorted_keys(keys, data, shape, x.fill_value, it)
    BROADCAST_STATS["general"] = BROADCAST_STATS.get("general", 0) + 1
    # base[e]: x's coordinates weighted by the TARGET strides of the axes x really has
    lead = nd - x.ndim
    base = torch.empty(nnz, dtype=torch.int64, device=dev)
    _ffi.call("spamd_coo_linearize", K.code_of(x.coords.dtype), x.ndim, nnz, ptr(x.coords.contiguous()), nnz,
              K._harr64(strides[lead:]), K._harr32(range(x.ndim)), ptr(base), stream_ptr(dev))
    # off[r]: host-side (rep entries) offsets of the replicas along the broadcast axes
    bshape = [shape[d] for d in bdims]
    grids = np.indices(bshape).reshape(len(bdims), -1) if bdims else np.zeros((0, 1), dtype=np.int64)
    off = np.zeros(rep, dtype=np.int64)
    for j, d in enumerate(bdims):
        off += grids[j].astype(np.int64) * strides[d]
    off_t = torch.from_numpy(off).to(dev)
    n_out = nnz * rep
    t = torch.empty(n_out, dtype=torch.int64, device=dev)
    _ffi.call("spamd_iota", n_out, ptr(t), stream_ptr(dev))
    rep_t = torch.tensor([rep], dtype=torch.int64, device=dev)
    e = binary_arrays("floor_divide_i64", t, rep_t, b_scalar=True)
    r = binary_arrays("subtract", t, binary_arrays("multiply", e, rep_t, b_scalar=True))
    keys = binary_arrays("add", K.gather(base, e), K.gather(off_t, r))
    data = K.gather(x.data, e)
    size = 1
    for s in shape:
        size *= s
    keys, perm = K.sort_keys(keys, max(size - 1, 1))
    data = K.gather(data, perm)
    it = x.coords.dtype if max(shape) < 2 ** 31 or x.coords.dtype == torch.int64 else torch.int64
    out = COO(K.delinearize(keys, shape, it), data, shape=shape, has_duplicates=False, sorted=True,
              fill_value=x.fill_value)
    out._keys = keys
    return out


def broadcast_pair(a, b):
    shape = broadcast_shapes(a.shape, b.shape)
    return broadcast_to(a, shape), broadcast_to(b, shape)
