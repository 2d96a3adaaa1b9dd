"""N-D matmul broadcasting and the stack/concatenate/leading-index helpers it needs
(reference `_matmul_recurser`, _common.py:278-293; `_coo/common.py:132-249`;
`_compressed/common.py:6-96`).

Everything stays on linear keys: `x[i]` is a contiguous key range of a canonical COO (found by
two binary searches), `stack`/`concatenate` offset the keys of each piece and append them (pieces
along axis 0 are already in order, so no re-sort)."""
import numpy as np
import torch

from . import _ffi
from . import _kernels as K
from ._device import ptr, stream_ptr
from ._utils import prod


def _key_range(keys, lo_key, hi_key):
    """[first index with key >= lo_key, first index with key >= hi_key) in a sorted key array."""
    dev = keys.device
    q = torch.tensor([lo_key, hi_key], dtype=torch.int64, device=dev)
    pos = torch.empty(2, dtype=torch.int64, device=dev)
    m = torch.empty(3, dtype=torch.int64, device=dev)
    _ffi.call("spamd_lower_bound_match", 2, ptr(q), keys.numel(), ptr(keys), ptr(pos), ptr(m), stream_ptr(dev))
    lo, hi = pos.tolist()
    return int(lo), int(hi)


def take_leading(x, i):
    """`x[i]` for an integer i on the leading axis (COO or GCXS; result has the same format)."""
    from ._coo import COO
    from ._gcxs import GCXS
    from ._umath import binary_arrays

    if x.ndim == 0:
        raise IndexError("too many indices for array")
    n0 = x.shape[0]
    if i < 0:
        i += n0
    if not 0 <= i < n0:
        raise IndexError(f"index {i} is out of bounds for axis 0 with size {n0}")
    is_gcxs = isinstance(x, GCXS)
    c = x.tocoo() if is_gcxs else x
    stride = 1
    for s in c.shape[1:]:
        stride *= s
    keys = c.linear_loc()
    lo, hi = _key_range(keys, i * stride, (i + 1) * stride)
    sub_keys = keys[lo:hi].contiguous()
    if hi > lo and i:
        sub_keys = binary_arrays("subtract", sub_keys, torch.tensor([i * stride], dtype=torch.int64, device=keys.device),
                                 b_scalar=True)
    out = COO(c.coords[1:, lo:hi].contiguous(), c.data[lo:hi].contiguous(), shape=c.shape[1:], has_duplicates=False,
              sorted=True, fill_value=c.fill_value)
    out._keys = sub_keys
    return out.asformat("gcxs") if is_gcxs and out.ndim >= 1 else out


def _gcxs_result(out, nd_in, axis, compressed_axes):
    """What the reference's all-GCXS branch returns (`_compressed/common.py:6-96`): 1-D inputs give a COO, otherwise a
    GCXS compressed along `compressed_axes` (default: the joined axis)."""
    if nd_in == 1:
        return out
    return out.asformat("gcxs", compressed_axes=(axis,) if compressed_axes is None else compressed_axes)


CONCAT_MERGE = True


def _concatenate_compressed(arrays, axis, compressed_axes):
    """Matrices compressed along the axis they are joined on (CSR matrices stacked by rows, CSC by columns): the value and
    index arrays one after the other, the pointers shifted by what came before - no conversion to COO and back (2.2 ms for
    two operands of 10^7 stored elements; the reference's all-GCXS branch, `_compressed/common.py:6-96`, goes through the
    pointers the same way).  None = not this case (the checks and their errors are the general route's)."""
    from ._gcxs import GCXS, unified_index_dtype
    from ._umath import binary_arrays

    a0 = arrays[0]
    if a0.ndim != 2 or not isinstance(axis, (int, np.integer)) or not -2 <= axis < 2:
        return None
    axis = int(axis) % 2
    if compressed_axes is not None and tuple(compressed_axes) != (axis,):
        return None
    other = a0.shape[1 - axis]
    for a in arrays:
        if a.ndim != 2 or a.compressed_axes != (axis,) or a.shape[1 - axis] != other or a.device != a0.device \
                or not np.array_equal(np.asarray(a.fill_value), np.asarray(a0.fill_value), equal_nan=True):
            return None
    dt = np.result_type(*[a.dtype for a in arrays])
    total_nnz = sum(int(a.data.numel()) for a in arrays)
    it = torch.int64 if any(a.indices.dtype == torch.int64 for a in arrays) else unified_index_dtype(torch.int32, total_nnz)
    ptrs, off = [torch.zeros(1, dtype=it, device=a0.device)], 0
    for a in arrays:
        p = a.indptr[1:].to(it) if a.indptr.dtype != it else a.indptr[1:].clone()
        if off and p.numel():
            p = binary_arrays("add", p, torch.tensor([off], dtype=it, device=p.device), b_scalar=True)
        ptrs.append(p)
        off += int(a.data.numel())
    n = sum(a.shape[axis] for a in arrays)
    shape = (n, other) if axis == 0 else (other, n)
    return GCXS((torch.cat([K.convert(a.data, dt) for a in arrays]), torch.cat([a.indices.to(it) for a in arrays]), torch.cat(ptrs)),
                shape=shape, compressed_axes=(axis,), fill_value=np.asarray(a0.fill_value).astype(dt)[()])


def concatenate(arrays, axis=0, compressed_axes=None):
    """Join sparse arrays along an existing axis (reference _common.py:1518-1554, _coo/common.py:132-192)."""
    from ._coo import COO, as_coo
    from ._gcxs import GCXS
    from ._umath import binary_arrays
    from ._utils import normalize_axis

    arrays = list(arrays)
    if not arrays:
        raise ValueError("need at least one array to concatenate")
    all_gcxs = all(isinstance(a, GCXS) for a in arrays)
    if all_gcxs:
        fast = _concatenate_compressed(arrays, axis, compressed_axes)
        if fast is not None:
            return fast
    coos = [as_coo(a) for a in arrays]
    fv = coos[0].fill_value
    for k, c in enumerate(coos):
        if not np.array_equal(np.asarray(c.fill_value), np.asarray(fv), equal_nan=True):
            raise ValueError("This operation requires consistent fill-values, "
                             f"but argument {k:d} had a fill value of {c.fill_value!s}, which "
                             f"is different from a fill_value of {fv!s} in the first argument.")
    axis = normalize_axis(axis, coos[0].ndim)
    ref_shape = coos[0].shape
    for c in coos:
        if c.ndim != len(ref_shape) or any(c.shape[d] != ref_shape[d] for d in range(c.ndim) if d != axis):
            raise ValueError("All arrays must have the same shape except for the concatenation axis.")
    dt = np.result_type(*[c.dtype for c in coos])
    dev = coos[0].device
    it = torch.int64 if any(c.coords.dtype == torch.int64 for c in coos) else torch.int32
    total = sum(c.shape[axis] for c in coos)
    shape = tuple(total if d == axis else ref_shape[d] for d in range(len(ref_shape)))
    parts_c, parts_d, off = [], [], 0
    for c in coos:
        cc = c.coords.to(it).clone()
        if off and c.nnz:
            cc[axis] = binary_arrays("add", cc[axis].contiguous(), torch.tensor([off], dtype=it, device=dev), b_scalar=True)
        parts_c.append(cc)
        parts_d.append(K.convert(c.data, dt))
        off += c.shape[axis]
    out = None
    if axis != 0 and CONCAT_MERGE and len(coos) <= 8 and all(p.dtype == parts_d[0].dtype for p in parts_d) \
            and parts_d[0].element_size() in (1, 4, 8) and prod(shape) < 2 ** 62:
        # joined along an inner axis, every operand's elements keep their order among themselves (canonical operands, one
        # coordinate shifted): the result is a MERGE of k sorted key arrays, not a sort of their concatenation (1.66 ms for two
        # operands of 10^7 stored elements, `tools/r06/index_sweep.py`)
        from ._umath import _as_u8, union_merge

        keys, data = None, None
        for cc, dd in zip(parts_c, parts_d):
            if not cc.shape[1]:
                continue
            k = K.linearize(cc, shape)
            if keys is None:
                keys, data = k, dd
                continue
            keys, sa, sb = union_merge(keys, k)
            nd = torch.empty(int(keys.numel()), dtype=dd.dtype, device=dev)
            K.scatter_into(_as_u8(nd), sa, _as_u8(data.contiguous()))
            K.scatter_into(_as_u8(nd), sb, _as_u8(dd.contiguous()))
            data = nd
        if keys is not None:
            out = COO._from_sorted_keys(keys, data, shape, np.asarray(fv).astype(dt)[()], it)
    if out is None:
        out = COO(torch.cat(parts_c, dim=1), torch.cat(parts_d), shape=shape, has_duplicates=False, sorted=(axis == 0),
                  fill_value=fv)
    return _gcxs_result(out, len(ref_shape), axis, compressed_axes) if all_gcxs else out


def stack(arrays, axis=0, compressed_axes=None):
    """Join sparse arrays along a NEW axis (reference _common.py:1479-1515, _coo/common.py:195-249)."""
    from ._coo import COO, as_coo
    from ._gcxs import GCXS

    arrays = list(arrays)
    if not arrays:
        raise ValueError("need at least one array to stack")
    all_gcxs = all(isinstance(a, GCXS) for a in arrays)
    coos = [as_coo(a) for a in arrays]
    if any(c.shape != coos[0].shape for c in coos):
        raise ValueError("All arrays must have the same shape.")
    nd = coos[0].ndim
    if axis < 0:
        axis += nd + 1
    lifted = [c[(None,)] for c in coos]              # new leading axis of length 1
    out = concatenate(lifted, axis=0)                 # stack along axis 0 ...
    if axis != 0:                                     # ... then move it into place
        perm = list(range(1, nd + 1))
        perm.insert(axis, 0)
        out = out.transpose(perm)
    return _gcxs_result(out, nd, axis, compressed_axes) if all_gcxs else out


def block_diagonal_csr(a):
    """The batch of matrices a[..., M, K] as ONE CSR matrix: block b sits at rows [b*M, (b+1)*M) and columns
    [b*K, (b+1)*K).  With row = key // K (rows of the stacked (B*M, K) view) the new linear key is
    row * (B*K) + (row // M) * K + key % K — monotone in the old key, so the canonical order is kept and the
    CSR arrays follow from one `keys_to_csr` pass."""
    from ._convert import _pick_index_dtype
    from ._coo import as_coo
    from ._gcxs import GCXS
    from ._umath import binary_arrays

    c = as_coo(a)
    M, Kd = c.shape[-2], c.shape[-1]
    B = 1
    for s in c.shape[:-2]:
        B *= s
    keys = c.linear_loc()
    dev = c.device

    def sc(v):
        return torch.tensor([v], dtype=torch.int64, device=dev)

    if c.nnz:
        row = binary_arrays("floor_divide_i64", keys, sc(max(Kd, 1)), b_scalar=True)
        col = binary_arrays("subtract", keys, binary_arrays("multiply", row, sc(Kd), b_scalar=True))
        blk = binary_arrays("floor_divide_i64", row, sc(max(M, 1)), b_scalar=True)
        keys = binary_arrays("add", binary_arrays("multiply", row, sc(B * Kd), b_scalar=True),
                             binary_arrays("add", binary_arrays("multiply", blk, sc(Kd), b_scalar=True), col))
    it = _pick_index_dtype(c._index_dtype, max(B * M, B * Kd, c.nnz))
    indptr, indices = K.keys_to_csr(keys, B * M, B * Kd, it)
    return GCXS((c.data, indices, indptr), shape=(B * M, B * Kd), compressed_axes=(0,), fill_value=c.fill_value)


def matmul_blockdiag(a, b):
    """Batched `a @ b` for a sparse `a[..., M, K]` and `b[..., K, N]` with the SAME leading axes, as a single 2-D
    product: blockdiag(a) (B*M x B*K) times b stacked to (B*K, N).  One kernel launch sequence for the whole batch
    (SpMM for a dense b, SpGEMM for a sparse b) instead of the reference's Python loop over slices
    (`_matmul_recurser`, _common.py:278-293); every output row still accumulates its own block in k order."""
    from ._dot import dot
    from ._sparse_array import SparseArray

    lead = tuple(a.shape[:-2])
    M, Kd, N = a.shape[-2], a.shape[-1], b.shape[-1]
    B = 1
    for s in lead:
        B *= s
    big = block_diagonal_csr(a)
    if isinstance(b, SparseArray):
        res = dot(big, b.reshape((B * Kd, N)))
        res = res.reshape(lead + (M, N))
        from ._gcxs import GCXS

        return res if isinstance(a, GCXS) and isinstance(b, GCXS) else res.asformat("coo")
    # a NumPy `b` goes through `dot` as an ndarray, so the result comes back as an ndarray (drop-in: NumPy in -> NumPy
    # out, like the 2-D path and the per-slice loop); a device tensor stays on the device
    bt = b if isinstance(b, torch.Tensor) else np.ascontiguousarray(b)
    res = dot(big, bt.reshape(B * Kd, N))
    return res.reshape(lead + (M, N))


def _prod(xs):
    n = 1
    for x in xs:
        n *= int(x)
    return n


def _swap_last(x):
    """x with its last two axes exchanged (sparse: a key permutation + sort; dense: a strided view made contiguous)"""
    from ._sparse_array import SparseArray

    nd = x.ndim
    perm = tuple(range(nd - 2)) + (nd - 1, nd - 2)
    if isinstance(x, SparseArray):
        return x.transpose(perm)
    if isinstance(x, torch.Tensor):
        return x.permute(perm).contiguous()
    return np.ascontiguousarray(np.transpose(x, perm))


def matmul_broadcast(a, b):
    """`a @ b` for N-D operands whose leading axes BROADCAST (sizes 1 against n), as one product instead of the
    reference's Python recursion over slices (`_matmul_recurser`, _common.py:278-293).  With the leading axes split into
    E (both operands have the axis), A (only `a` does, `b` has 1) and B (only `b` does):

        out[e, x, y, m, n] = sum_k a[e, x, m, k] * b[e, y, k, n]
                           = ( a viewed as [E, (A*M), K] )  @  ( b viewed as [E, K, (B*N)] )      per e,

    i.e. an equal-leading-axes batch (`matmul_blockdiag`, or a plain 2-D product when E is empty) of taller / wider
    matrices, followed by a reshape to [E, A, M, B, N] and a transposition into the broadcast order.  The views are key
    permutations for sparse operands and strided copies for dense ones; every output element still sums its own k terms
    in k order.  `a` dense and `b` sparse goes through (b^T a^T)^T."""
    from ._dot import dot
    from ._gcxs import GCXS
    from ._sparse_array import SparseArray

    if not isinstance(a, SparseArray):
        r = matmul_broadcast(_swap_last(b), _swap_last(a))
        return _swap_last(r)
    nd = a.ndim
    la, lb = tuple(a.shape[:-2]), tuple(b.shape[:-2])
    M, Kd, N = int(a.shape[-2]), int(a.shape[-1]), int(b.shape[-1])
    E = [d for d in range(nd - 2) if la[d] == lb[d]]
    A = [d for d in range(nd - 2) if la[d] != lb[d] and lb[d] == 1]
    B = [d for d in range(nd - 2) if la[d] != lb[d] and la[d] == 1]
    eshape, ashape, bshape = [la[d] for d in E], [la[d] for d in A], [lb[d] for d in B]
    nA, nB = _prod(ashape), _prod(bshape)
    b_sparse = isinstance(b, SparseArray)

    # a -> [E..., A..., M, K] (its B axes have length 1: dropped by the reshape), then [E..., A*M, K]
    perm_a = E + A + B + [nd - 2, nd - 1]
    a2 = a if perm_a == list(range(nd)) else a.transpose(perm_a)
    a2 = a2.reshape(tuple(eshape) + (nA * M, Kd))
    # b -> [E..., K, B..., N] (its A axes have length 1), then [E..., K, B*N]
    perm_b = E + A + [nd - 2] + B + [nd - 1]
    if b_sparse:
        b2 = b if perm_b == list(range(nd)) else b.transpose(perm_b)
        b2 = b2.reshape(tuple(eshape) + (Kd, nB * N))
    else:
        was_numpy = not isinstance(b, torch.Tensor)
        bt = torch.from_numpy(np.ascontiguousarray(b)) if was_numpy else b
        b2 = bt.permute(perm_b).reshape(tuple(eshape) + (Kd, nB * N))
        b2 = np.ascontiguousarray(b2.numpy()) if was_numpy else b2.contiguous()
    res = matmul_blockdiag(a2, b2) if eshape else dot(a2, b2)
    # [E..., A*M, B*N] -> [E..., A..., M, B..., N] -> broadcast order [lead..., M, N]
    mid = tuple(eshape) + tuple(ashape) + (M,) + tuple(bshape) + (N,)
    pos, src = {}, 0
    for d in E:
        pos[d] = src
        src += 1
    for d in A:
        pos[d] = src
        src += 1
    m_pos = src
    src += 1
    for d in B:
        pos[d] = src
        src += 1
    n_pos = src
    back = [pos[d] for d in range(nd - 2)] + [m_pos, n_pos]
    if isinstance(res, SparseArray):
        res = res.reshape(mid)
        if back != list(range(len(mid))):
            res = res.transpose(back)
        return res if isinstance(a, GCXS) and isinstance(b, GCXS) else res.asformat("coo")
    if isinstance(res, torch.Tensor):
        return res.reshape(mid).permute(back).contiguous()
    return np.ascontiguousarray(np.transpose(res.reshape(mid), back))


def matmul_batched(a, b):
    """`_matmul_recurser` (reference _common.py:278-293): loop over the broadcast leading axis,
    2-D `dot` per slice, stack the results."""
    from ._dot import dot
    from ._sparse_array import SparseArray

    def idx(x, i):
        if isinstance(x, SparseArray):
            return x[i]
        return x[i]

    def rec(a, b):
        if a.ndim == 2:
            return dot(a, b)
        res = []
        for i in range(max(a.shape[0], b.shape[0])):
            a_i = idx(a, 0) if a.shape[0] == 1 else idx(a, i)
            b_i = idx(b, 0) if b.shape[0] == 1 else idx(b, i)
            res.append(rec(a_i, b_i))
        if all(isinstance(x, SparseArray) for x in res):
            return stack(res)
        res = [x.todense_device() if isinstance(x, SparseArray) else x for x in res]
        if all(isinstance(x, np.ndarray) for x in res):
            return np.stack(res)
        dev0 = next(x.device for x in res if isinstance(x, torch.Tensor))
        return torch.stack([x if isinstance(x, torch.Tensor) else torch.from_numpy(x).to(dev0) for x in res])

    return rec(a, b)
