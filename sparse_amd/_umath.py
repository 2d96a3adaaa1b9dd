"""A7: `elemwise` — N-ary elementwise functions over sparse operands (reference
sparse/numba_backend/_umath.py:13-50,392-751).

The reference handles arbitrary Python callables by enumerating presence masks and joining
coordinates per mask.  On canonical operands that collapses to a sorted-key union (see
csrc/ewise.hip); this module implements that HIP path for NumPy ufuncs over
{sparse (x) sparse (same shape), sparse (x) scalar, sparse (x) same-shape dense, unary}, with the
reference's fill-value rules (result fill = func(fills); stored results bit-equal to it are
pruned; non-zero operand fills are honoured).  Anything else raises NotImplementedError —
there is deliberately no host fallback.
"""
import threading

import numpy as np
import torch

from . import _device as dev
from . import _ffi
from . import _kernels as K
from ._device import ptr, require_hip, stream_ptr, torch_dtype

# NumPy ufunc name -> (kind, C-ABI op code)   (codes: include/sparse_amd.h)
_BIN = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "true_divide": 3, "maximum": 4, "minimum": 5,
        "power": 6, "fmax": 7, "fmin": 8, "greater": 32, "greater_equal": 33, "less": 34, "less_equal": 35,
        "equal": 36, "not_equal": 37, "logical_and": 38, "logical_or": 39, "logical_xor": 40,
        "bitwise_and": 64, "bitwise_or": 65, "bitwise_xor": 66,
        # late round 6 (tools/r06/ufunc_sweep.py: 70-290 ms per call on the host at 10^7 stored elements): value kernels only,
        # not in the fused merge (`_UNFUSED`)
        "float_power": 6, "floor_divide": 9, "remainder": 10, "mod": 10, "fmod": 11, "copysign": 12, "hypot": 13, "arctan2": 14,
        "left_shift": 67, "right_shift": 68}
_UNFUSED = {6, 9, 10, 11, 12, 13, 14, 67, 68}       # ops csrc/merge.hip does not evaluate: union first, then the value kernel
_F = (np.dtype("f4"), np.dtype("f8"))
_FI = _F + (np.dtype("i4"), np.dtype("i8"))
_COMP_TYPES = {9: _FI, 10: _FI, 11: _FI, 12: _F, 13: _F, 14: _F, 67: _FI[2:], 68: _FI[2:]}      # compute types of the late ops
_UN = {"negative": 0, "absolute": 1, "abs": 1, "fabs": 1, "sqrt": 2, "exp": 3, "expm1": 4, "log": 5, "log1p": 6,
       "sin": 7, "cos": 8, "tan": 9, "tanh": 10, "sinh": 11, "cosh": 12, "arcsin": 13, "arctan": 14, "floor": 15,
       "ceil": 16, "rint": 17, "trunc": 18, "sign": 19, "square": 20, "reciprocal": 21, "positive": 22, "log2": 23,
       "log10": 24, "exp2": 25, "arcsinh": 26, "arctanh": 27, "cbrt": 28, "deg2rad": 29, "radians": 29,
       "rad2deg": 30, "degrees": 30, "isnan": 64, "isinf": 65, "isfinite": 66, "logical_not": 67, "signbit": 68,
       "conjugate": 22, "conj": 22, "real": 22}
_TO_BOOL_BIN = set(range(32, 41))
_BOOL_ARITH = {"add": "logical_or", "maximum": "logical_or", "fmax": "logical_or",
               "multiply": "logical_and", "minimum": "logical_and", "fmin": "logical_and"}
_CODE = {torch.float32: _ffi.F32, torch.float64: _ffi.F64, torch.int32: _ffi.I32, torch.int64: _ffi.I64,
         torch.uint8: _ffi.U8, torch.bool: _ffi.U8}


def _as_u8(t):
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def binary_arrays(name, a, b, a_scalar=False, b_scalar=False, out_bool_as=torch.bool):
    """out = a (op) b on equal-length device arrays (or a 1-element array broadcast as scalar)."""
    if name == "floor_divide_i64":  # non-negative int64 keys: truncation == floor
        code, name = 3, "divide"
    else:
        code = _BIN[name]
    devi = require_hip(a, b)
    a, b = _as_u8(a.contiguous()), _as_u8(b.contiguous())
    if a.dtype != b.dtype:
        raise TypeError(f"binary_arrays needs one compute dtype, got {a.dtype} and {b.dtype}")
    n = int(b.numel() if a_scalar else a.numel())
    out_dtype = torch.uint8 if code in _TO_BOOL_BIN else a.dtype
    out = torch.empty(n, dtype=out_dtype, device=devi)
    _ffi.call("spamd_ewise_binary", code, _CODE[a.dtype], n, ptr(a), int(a_scalar), ptr(b), int(b_scalar), ptr(out),
              stream_ptr(devi))
    return out.view(torch.bool) if (code in _TO_BOOL_BIN and out_bool_as == torch.bool) else out


def unary_array(name, a):
    code = _UN[name]
    devi = require_hip(a)
    a = _as_u8(a.contiguous())
    out = torch.empty(a.numel(), dtype=torch.uint8 if code >= 64 else a.dtype, device=devi)
    _ffi.call("spamd_ewise_unary", code, _CODE[a.dtype], a.numel(), ptr(a), ptr(out), stream_ptr(devi))
    return out.view(torch.bool) if code >= 64 else out


def select(mask, a, b):
    """where(mask, a, b) on device arrays, built from flag/compact primitives: scatter-free form
    out = b; out[mask] = a[mask]."""
    devi = require_hip(mask, a, b)
    m = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    if m.dtype == torch.uint8 and a.dtype == b.dtype and a.element_size() in (1, 4, 8) and a.numel() == b.numel() == m.numel():
        # one pass (`spamd_ewise_select`: 0/1 mask bytes, values moved bit-wise) instead of clone + flags + scan + iota +
        # compact + gather + scatter with a host read in the middle (late round 6: where(x > 0.5, x, 0) at 10^7 stored
        # elements 1.51 ms, half of it here)
        out = torch.empty_like(b.contiguous())
        _ffi.call("spamd_ewise_select", out.element_size(), int(out.numel()), ptr(m.contiguous()), ptr(a.contiguous()), 0,
                  ptr(b.contiguous()), 0, ptr(out), stream_ptr(devi))
        return out
    out = b.clone()
    flags = K.flag_ne_bits(m, 0)
    offs = K.exclusive_scan(flags)
    cnt = int(offs[-1])
    if cnt:
        n = int(m.numel())
        iota = torch.empty(n, dtype=torch.int64, device=devi)
        _ffi.call("spamd_iota", n, ptr(iota), stream_ptr(devi))
        idx = K.compact(iota, flags, offs, cnt)
        K.scatter_into(out, idx, K.gather(a, idx))
    return out


def _full(n, value, dtype, devi):
    t = torch.empty(n, dtype=dtype, device=devi)
    if n:
        npdt = dev.np_dtype(dtype) if dtype != torch.bool else np.dtype("uint8")
        bits = int(np.asarray(value, dtype=npdt).reshape(1).view(f"u{npdt.itemsize}")[0])
        _ffi.call("spamd_fill", t.element_size(), n, ptr(t), bits, stream_ptr(devi))
    return t


def union_merge(ka, kb):
    """Sorted-key union of two canonical key arrays: (keys, slotA, slotB)."""
    devi = require_hip(ka, kb)
    na, nb = int(ka.numel()), int(kb.numel())
    posB = torch.empty(na, dtype=torch.int64, device=devi)
    mA = torch.empty(na + 1, dtype=torch.int64, device=devi)
    posA = torch.empty(nb, dtype=torch.int64, device=devi)
    mB = torch.empty(nb + 1, dtype=torch.int64, device=devi)
    s = stream_ptr(devi)
    _ffi.call("spamd_lower_bound_match", na, ptr(ka), nb, ptr(kb), ptr(posB), ptr(mA), s)
    _ffi.call("spamd_lower_bound_match", nb, ptr(kb), na, ptr(ka), ptr(posA), ptr(mB), s)
    um = torch.empty(nb + 1, dtype=torch.int64, device=devi)
    _ffi.call("spamd_invert_flags", nb, ptr(mB), ptr(um), s)
    ub = K.exclusive_scan(um)
    n_out = na + int(ub[-1])
    slotA = torch.empty(na, dtype=torch.int64, device=devi)
    slotB = torch.empty(nb, dtype=torch.int64, device=devi)
    keys = torch.empty(n_out, dtype=torch.int64, device=devi)
    _ffi.call("spamd_union_positions", na, ptr(ka), ptr(posB), nb, ptr(kb), ptr(posA), ptr(mB), ptr(ub), ptr(slotA),
              ptr(slotB), ptr(keys), s)
    return keys, slotA, slotB


def _bits(value, np_dtype):
    npdt = np.dtype("uint8") if np.dtype(np_dtype) == np.dtype(bool) else np.dtype(np_dtype)
    return int(np.asarray(value).astype(npdt).reshape(1).view(f"u{npdt.itemsize}")[0])


MERGE_SINGLE_PASS = True   # tuning hook: False = count pass + scan + fill pass
MERGE_FUSED = True         # one launch (partition inside the kernel, self-cleaning workspace, total through pinned memory)
# ... up to this many items: every tile of the fused form searches its own two merge-path diagonals with 64 probes per round
# (latency: 4 rounds instead of ~23), which touches ~0.5 K cache lines per tile.  At config 1 (2 x 10^6 items, everything
# resident in the Infinity Cache) that is the cheaper trade: 0.109 -> 0.062 ms per `x + y`; at 2 x 10^8 items the probes
# are HBM traffic of the size of the operands themselves (2.4 -> 4.5 ms), so large merges keep the partition kernel.
# Late round 4 (1024-thread tiles; tools/r04/merge_fused_crossover.py, fused / partition + single pass, ms): 2^21 items 0.050 / 0.061,
# 2^22 0.083 / 0.080, 2^23 0.148 / 0.112, 2^24 0.305 / 0.199 - the bound moves from 2^23 to 2^22.
MERGE_FUSED_MAX_ITEMS = 1 << 22


class _MergeWorkspace:
    """Per (device, stream): the fused merge kernel's look-back workspace - zero before the first call, left zero by every
    call - plus one device word and one pinned host word for the number of outputs.  The host does not copy the count
    back: the kernel stores it into the pinned word as soon as the last tile knows it and the host spins on that word
    (a `.item()` costs a blocking stream synchronisation plus a copy, ~20 us; config-1-sized merges take ~35 us)."""

    _pool = {}
    SENTINEL = -1
    SLOTS = 64      # pinned result words used round-robin: the late store of a call that was abandoned (an interrupt in
    #                 the spin, a fault) lands in ITS slot, not in the word the next call is waiting on

    def __init__(self, devi):
        self.cap = 0
        self.ws = None
        self.total_dev = torch.zeros(1, dtype=torch.int64, device=devi)
        self.pinned = torch.full((self.SLOTS,), self.SENTINEL, dtype=torch.int64).pin_memory()
        self.view = self.pinned.numpy()
        self.devi = devi
        self.seq = 0
        self.lock = threading.Lock()    # two host threads on one stream share the workspace: one merge at a time

    def next_slot(self):
        self.seq += 1
        k = self.seq % self.SLOTS
        self.view[k] = self.SENTINEL
        return k

    def recover(self):
        """After a failed or abandoned call: wait for whatever is in flight and restore the all-zero workspace the next
        call relies on (the kernel cleans up after itself only when it runs to completion)."""
        try:
            torch.cuda.current_stream(self.devi).synchronize()
        finally:
            if self.ws is not None:
                self.ws.zero_()

    @classmethod
    def get(cls, devi, stream, nblocks):
        key = (devi.index, stream)
        w = cls._pool.get(key)
        if w is None:
            w = cls._pool[key] = cls(devi)
        if nblocks > w.cap:
            w.cap = max(2 * nblocks, 4096)
            w.ws = torch.zeros(3 + w.cap, dtype=torch.int64, device=devi)
        return w

    def wait_total(self, k):
        """Spin on pinned word `k` (bounded: ~50 ms), then fall back to the device word behind a synchronisation."""
        view = self.view
        for _ in range(2_000_000):
            t = int(view[k])
            if t != self.SENTINEL:
                return t
        torch.cuda.current_stream(self.devi).synchronize()
        t = int(view[k])
        return t if t != self.SENTINEL else int(self.total_dev[0])


def merge_union(name, ka, va, kb, vb, fill_a, fill_b, fill_out):
    """Fused merge-path union: (keys, values) of func(a or fill_a, b or fill_b) over the union of
    two canonical key arrays, results bit-equal to `fill_out` dropped (csrc/merge.hip).
    `va`/`vb` share the compute dtype; `fill_out` is a NumPy scalar of the output dtype."""
    code = _BIN[name]
    devi = require_hip(ka, kb, va, vb)
    va, vb = _as_u8(va.contiguous()), _as_u8(vb.contiguous())
    comp_np = dev.np_dtype(va.dtype) if va.dtype != torch.uint8 else np.dtype("uint8")
    na, nb = int(ka.numel()), int(kb.numel())
    out_t = torch.uint8 if code in _TO_BOOL_BIN else va.dtype
    nblocks = int(_ffi.lib().spamd_merge_num_blocks(na, nb))
    if nblocks == 0:
        e = torch.empty(0, dtype=out_t, device=devi)
        return torch.empty(0, dtype=torch.int64, device=devi), (e.view(torch.bool) if code in _TO_BOOL_BIN else e)
    s = stream_ptr(devi)
    fa, fb = _bits(fill_a, comp_np), _bits(fill_b, comp_np)
    fo = _bits(fill_out, np.dtype("uint8") if code in _TO_BOOL_BIN else comp_np)
    if MERGE_FUSED and MERGE_SINGLE_PASS and na + nb <= MERGE_FUSED_MAX_ITEMS:
        w = _MergeWorkspace.get(devi, s, int(_ffi.lib().spamd_merge_fused_blocks(na, nb)))
        keys = torch.empty(na + nb, dtype=torch.int64, device=devi)
        vals = torch.empty(na + nb, dtype=out_t, device=devi)
        with w.lock:
            k = w.next_slot()
            try:
                _ffi.call("spamd_merge_union_fused", code, _CODE[va.dtype], na, ptr(ka), ptr(va), nb, ptr(kb), ptr(vb), fa, fb,
                          fo, ptr(w.ws), ptr(w.total_dev), w.pinned.data_ptr() + 8 * k, ptr(keys), ptr(vals), s)
                total = w.wait_total(k)
            except BaseException:      # KeyboardInterrupt in the spin included: leave the workspace as the next call needs it
                w.recover()
                raise
        if total * 2 < na + nb:   # a sparse result should not pin the worst-case buffers
            keys, vals = keys[:total].clone(), vals[:total].clone()
        else:
            keys, vals = keys[:total], vals[:total]
        return keys, (vals.view(torch.bool) if code in _TO_BOOL_BIN else vals)
    part = torch.empty(nblocks + 1, dtype=torch.int64, device=devi)
    _ffi.call("spamd_merge_partition", na, ptr(ka), nb, ptr(kb), ptr(part), s)
    counts = torch.empty(nblocks + 2, dtype=torch.int64, device=devi)
    args = (code, _CODE[va.dtype], na, ptr(ka), ptr(va), nb, ptr(kb), ptr(vb), fa, fb, fo, ptr(part))
    if MERGE_SINGLE_PASS:
        # one pass: tiles chain their output offsets by look-back; room for the worst case, trimmed afterwards
        keys = torch.empty(na + nb, dtype=torch.int64, device=devi)
        vals = torch.empty(na + nb, dtype=out_t, device=devi)
        _ffi.call("spamd_merge_union", 2, *args, ptr(counts), 0, ptr(keys), ptr(vals), s)
        total = int(counts[nblocks + 1])
        if total * 2 < na + nb:   # a sparse result should not pin the worst-case buffers
            keys, vals = keys[:total].clone(), vals[:total].clone()
        else:
            keys, vals = keys[:total], vals[:total]
        return keys, (vals.view(torch.bool) if code in _TO_BOOL_BIN else vals)
    _ffi.call("spamd_merge_union", 0, *args, ptr(counts), 0, 0, 0, s)
    offs = K.exclusive_scan(counts[:nblocks + 1])   # n counts + one ignored slot (the last slot is single-pass only)
    total = int(offs[-1])
    keys = torch.empty(total, dtype=torch.int64, device=devi)
    vals = torch.empty(total, dtype=out_t, device=devi)
    _ffi.call("spamd_merge_union", 1, *args, 0, ptr(offs), ptr(keys), ptr(vals), s)
    return keys, (vals.view(torch.bool) if code in _TO_BOOL_BIN else vals)


def _where(proc, finish):
    """`np.where(condition, x, y)` over COO / scalar operands (reference `where`, _coo/common.py:534-581, which is
    `elemwise(np.where, ...)`): one sorted-key union of the sparse operands, each operand materialised on the union
    (stored value or its fill value), a device select, and the usual prune against where(fills)."""
    from ._broadcast import broadcast_shapes, broadcast_to
    from ._coo import COO

    sparse = [v for v in proc if isinstance(v, COO)]
    if any(not isinstance(v, COO) and not _scalar_like(v) for v in proc):
        raise NotImplementedError("where() with a dense (non-scalar) operand is not on the hip backend's path")
    shape = broadcast_shapes(*[v.shape for v in sparse])
    ops = [broadcast_to(v, shape) if isinstance(v, COO) else v for v in proc]
    devi = sparse[0].device

    def scalar_of(v):
        return np.asarray(v.item() if isinstance(v, torch.Tensor) else v)[()]

    def like(v):  # dtype-carrying stand-in; Python scalars stay weak (NEP 50)
        if isinstance(v, COO):
            return np.zeros(1, dtype=v.dtype)
        return v.item() if isinstance(v, torch.Tensor) else v

    fills = [np.asarray(v.fill_value)[()] if isinstance(v, COO) else scalar_of(v) for v in ops]
    out_np = _np_result(np.where, np.ones(1, dtype=bool), like(ops[1]), like(ops[2])).dtype
    if out_np not in (np.dtype("f4"), np.dtype("f8"), np.dtype("i4"), np.dtype("i8"), np.dtype("bool")):
        raise NotImplementedError(f"dtype {out_np} is not supported by the hip backend's where path")
    out_t = torch_dtype(out_np)
    fill = np.asarray(np.where(bool(fills[0]), fills[1], fills[2])).astype(out_np)[()]
    # union of the stored positions, with every operand's slots in it
    keys, slots = None, []
    for v in ops:
        if not isinstance(v, COO):
            slots.append(None)
            continue
        k = v.linear_loc()
        if keys is None:
            keys = k
            n = int(k.numel())
            iota = torch.empty(n, dtype=torch.int64, device=devi)
            if n:
                _ffi.call("spamd_iota", n, ptr(iota), stream_ptr(devi))
            slots.append(iota)
        else:
            keys, s_old, s_new = union_merge(keys, k)
            slots = [None if s is None else K.gather(s_old, s) for s in slots]
            slots.append(s_new)
    n = int(keys.numel())

    def on_union(v, slot, fv, dtype):
        fvn = np.asarray(fv).astype(dev.np_dtype(dtype) if dtype != torch.bool else bool)
        full = _full(n, fvn if dtype != torch.bool else np.uint8(bool(fvn)), torch.uint8 if dtype == torch.bool else dtype, devi)
        if isinstance(v, COO) and v.nnz:
            K.scatter_into(full, slot, _as_u8(K.convert(v.data, dtype)))
        return full

    cond = on_union(ops[0], slots[0], bool(fills[0]), torch.bool)
    xv = on_union(ops[1], slots[1], fills[1], out_t)
    yv = on_union(ops[2], slots[2], fills[2], out_t)
    res = select(cond, xv, yv) if n else xv
    if out_t == torch.bool:
        res = res.view(torch.bool)
    return finish(keys, res, shape, fill, devi)


def _union_slots(ops, devi):
    """Sorted-key union of the stored positions of the COO operands in `ops` (all of one shape):
    (keys, slots) with slots[i] = position of operand i's stored elements in the union (None for non-COO)."""
    from ._coo import COO

    keys, slots = None, []
    for v in ops:
        if not isinstance(v, COO):
            slots.append(None)
            continue
        k = v.linear_loc()
        if keys is None:
            keys = k
            n = int(k.numel())
            iota = torch.empty(n, dtype=torch.int64, device=devi)
            if n:
                _ffi.call("spamd_iota", n, ptr(iota), stream_ptr(devi))
            slots.append(iota)
        else:
            keys, s_old, s_new = union_merge(keys, k)
            slots = [None if s is None else K.gather(s_old, s) for s in slots]
            slots.append(s_new)
    return keys, slots


def _loose_all_equal(value, array):
    """reference `equivalent(value, array, loose=True).all()` (_utils.py:406-452): == or both NaN"""
    with np.errstate(all="ignore"):
        value, array = np.asarray(value), np.asarray(array)
        eq = value == array
        if value.dtype.kind in "fc" or array.dtype.kind in "fc":
            eq = eq | (np.isnan(value) & np.isnan(array))
        return bool(np.all(eq))


ELEMWISE_ON_DEVICE = True   # plain callables whose operations are exactly reproducible run on the device (`_trace`)
DEVICE_FALLBACKS = {"untraceable": 0, "declined": 0, "not_traced": 0}   # callables evaluated by NumPy on the host over the device-built union, by reason
_FALLBACK_LOG = []          # the last few (reason, function name, detail)


def _note_fallback(reason, func, detail=""):
    DEVICE_FALLBACKS[reason] = DEVICE_FALLBACKS.get(reason, 0) + 1
    _FALLBACK_LOG.append((reason, getattr(func, "__name__", repr(func)), detail))
    del _FALLBACK_LOG[:-32]


def fallback_stats(reset=False):
    """How often an elementwise callable left the device since import (or the last reset): {"untraceable": the tracer met
    an operation it has no exactly-rounded device kernel for, "declined": a kernel declined the dtype, "not_traced": the
    callable could not be turned into a graph at all (data-dependent control flow, indexing, an exception under symbolic
    operands), "recent": the last (reason, function, detail) triples}.  Those calls ARE correct - NumPy evaluates the
    function on the host over the union the device built - but cost a device -> host -> device round trip of
    every stored element: a count that grows in a hot loop is the thing to look at."""
    out = dict(DEVICE_FALLBACKS, recent=list(_FALLBACK_LOG))
    if reset:
        for k in DEVICE_FALLBACKS:
            DEVICE_FALLBACKS[k] = 0
        del _FALLBACK_LOG[:]
    return out


def _on_device(func, ops, slots, keys, n, full_shape, out_dtype, devi):
    """`func` over the union positions WITHOUT the nnz-sized host round trip: the callable is traced once into a graph of
    exactly-rounded elementwise operations (`_trace.build`) and replayed on device arrays (stored value or fill value per
    operand and union position; dense operands gathered at the union's positions).  None when `func` is not traceable -
    the caller then evaluates it with NumPy on the host, as the reference does."""
    from . import _trace
    from ._coo import COO

    spec = []
    for v in ops:
        if isinstance(v, COO):
            spec.append(("array", v.dtype))
        elif isinstance(v, np.ndarray) and v.ndim:
            spec.append(("array", v.dtype))
        else:
            spec.append(("scalar", v[()] if isinstance(v, np.ndarray) else v))
    root = _trace.build(func, spec)
    if root is None:
        _note_fallback("not_traced", func)
        return None
    arrays = []
    for v, slot in zip(ops, slots):
        if isinstance(v, COO):
            t = _full(n, np.asarray(v.fill_value)[()], v.data.dtype, devi)
            if v.nnz:
                K.scatter_into(_as_u8(t), slot, _as_u8(v.data))
            arrays.append(t)
        elif isinstance(v, np.ndarray) and v.ndim:
            flat = dev.to_device(np.ascontiguousarray(np.broadcast_to(v, full_shape)).reshape(-1), devi)
            arrays.append(K.gather(_as_u8(flat), keys).view(flat.dtype) if flat.dtype == torch.bool else K.gather(flat, keys))
        else:
            arrays.append(None)
    try:
        res = _trace.run(root, arrays, n, devi)
    except _trace.Untraceable as e:
        # ONLY the tracer's own verdict (an operation / dtype pair without a device kernel, a scalar that does not fit the
        # compute dtype): a KeyError / TypeError from the replay is a bug of this package and surfaces as one - behind a
        # broad `except` it would be a silent 100x slowdown (round-5 verdict)
        _note_fallback("untraceable", func, str(e))
        return None
    except _ffi.HipBackendError as e:
        if getattr(e, "code", 1) > 0:
            raise       # a hipError_t is a device fault, not "this kernel declines the dtype": never masked by the host path
        _note_fallback("declined", func, str(e))
        return None
    if res.dtype != torch_dtype(out_dtype):
        res = K.convert(res, torch_dtype(out_dtype))
    return res


def _elemwise_general(func, proc, kwargs, dtype_kw, finish):
    """The reference's `_Elemwise` for everything the fused kernels do not cover (_umath.py:392-751): ANY callable, any
    number of operands, dense operands, keyword arguments, non-zero fill values, broadcasting.
    The structure is built on the device (one sorted-key union of the stored positions instead of the reference's 2^k - 1
    mask enumeration + re-sort; every operand contributes its stored value or its fill value at each union position,
    which is what the masks evaluate coordinate by coordinate).  `func` itself is arbitrary Python: it is evaluated by
    NumPy on the host over nnz-sized arrays - the host fallback SURVEY.md section 7 (hard part 4) prescribes - with the
    reference's exceptions (`ValueError` on a densifying mixed operation, :541-546) and its dtype protocol
    (`dtype=` keyword when the function takes it, :617-625)."""
    from ._broadcast import broadcast_shapes, broadcast_to
    from ._coo import COO

    host = [dev.to_numpy(v) if isinstance(v, torch.Tensor) else v for v in proc]   # dense device tensors act as ndarrays
    devi = next(v for v in host if isinstance(v, COO)).device

    def shp(v):
        return tuple(v.shape) if isinstance(v, (COO, np.ndarray)) else ()

    full_shape = broadcast_shapes(*[shp(v) for v in host])
    ndarray_shape = broadcast_shapes(*[shp(v) for v in host if isinstance(v, np.ndarray)])

    def call(args, dtype):
        with np.errstate(all="ignore"):
            try:
                return func(*args, dtype=dtype, **kwargs)
            except TypeError:
                return func(*args, **kwargs)

    # ---- fill value (reference _get_fill_value, :505-555)
    zero_args = tuple(np.atleast_1d(np.asarray(v.fill_value)) if isinstance(v, COO)
                      else (np.atleast_1d(v) if isinstance(v, (np.generic, np.ndarray)) else v) for v in host)
    fill_array = np.asarray(call(zero_args, dtype_kw))
    fill = fill_array[(0,) * fill_array.ndim] if fill_array.size else \
        np.asarray(call(tuple(v.fill_value if isinstance(v, COO) else np.zeros((), np.asarray(v).dtype)[()] for v in host), None))[()]
    constant = _loose_all_equal(fill, fill_array)
    if not constant and full_shape != ndarray_shape:
        raise ValueError("Performing a mixed sparse-dense operation that would result in a dense array. "
                         "Please make sure that func(sparse_fill_values, ndarrays) is a constant array.")
    if not constant:   # same shapes: the reference densifies the sparse operands and returns func's dense result (:463-465)
        with np.errstate(all="ignore"):
            return func(*[v.todense() if isinstance(v, COO) else v for v in host], **kwargs)
    if dtype_kw is not None:
        fill = np.asarray(fill).astype(dtype_kw)[()]
    out_dtype = np.asarray(fill).dtype
    if any(d == 0 for d in full_shape):
        return finish(torch.empty(0, dtype=torch.int64, device=devi), torch.empty(0, dtype=torch_dtype(out_dtype), device=devi),
                      full_shape, fill, devi)

    # ---- values on the union of the stored positions
    ops = [broadcast_to(v, full_shape) if isinstance(v, COO) and tuple(v.shape) != tuple(full_shape) else v for v in host]
    keys, slots = _union_slots(ops, devi)
    n = int(keys.numel())
    if ELEMWISE_ON_DEVICE and dtype_kw is None and not kwargs and not isinstance(func, np.ufunc) and n:
        res = _on_device(func, ops, slots, keys, n, full_shape, out_dtype, devi)
        if res is not None:
            return finish(keys, res, full_shape, np.asarray(fill)[()], devi)
    hkeys = dev.to_numpy(keys)
    vals = []
    for v, slot in zip(ops, slots):
        if isinstance(v, COO):
            arr = np.full(n, np.asarray(v.fill_value)[()], dtype=v.dtype)
            if v.nnz:
                arr[dev.to_numpy(slot)] = dev.to_numpy(v.data)
            vals.append(arr)
        elif isinstance(v, np.ndarray) and v.ndim:
            vals.append(np.broadcast_to(v, full_shape).reshape(-1)[hkeys])
        else:
            vals.append(v)
    with np.errstate(all="ignore"):
        try:
            res = func(*vals, dtype=out_dtype, **kwargs)
        except TypeError:
            res = np.asarray(func(*vals, **kwargs)).astype(out_dtype)
    res = np.ascontiguousarray(np.broadcast_to(np.asarray(res).astype(out_dtype, copy=False), (n,)))
    return finish(keys, torch.from_numpy(res).to(devi), full_shape, np.asarray(fill)[()], devi)


def _broadcast_matched(name, func, a, b, shape, out_np, comp_np, finish):
    """`func(a, b)` where ONE operand (call it x) already has the broadcast `shape` and the other one (y) broadcasts into
    it, without materialising y's replicas: valid when the positions where x stores nothing can only produce the result's
    fill value, i.e. func(fill_x, v) is bit-equal to the result fill for EVERY stored v of y and for fill_y (multiply with
    zero-fill operands is the common case; checked on the device in one pass over y's nnz values).  Then the stored
    positions of the result are a subset of x's: every stored element of x looks up y at its coordinates PROJECTED onto
    y's axes (binary search in y's sorted keys; y's fill value where y stores nothing), func is applied and the result
    pruned - the reference's reduced-coordinate matching (`_get_matching_coords`, _umath.py:310-341) as a gather.
    Returns None when the condition does not hold (the caller then broadcasts by materialisation)."""
    x_is_a = tuple(a.shape) == tuple(shape)
    x, y = (a, b) if x_is_a else (b, a)
    devi = x.device
    comp_t = torch_dtype(comp_np)
    if comp_np not in (np.dtype("f4"), np.dtype("f8"), np.dtype("i4"), np.dtype("i8")) or np.dtype(out_np) != np.dtype(comp_np):
        return None
    fx, fy = np.asarray(x.fill_value).astype(comp_np), np.asarray(y.fill_value).astype(comp_np)
    fill = np.asarray(_np_result(func, *((fx, fy) if x_is_a else (fy, fx)))).astype(out_np)[()]

    def scalar_t(v):
        return torch.tensor([v.item() if hasattr(v, "item") else v], dtype=comp_t, device=devi)

    yd = K.convert(y.data, comp_t)
    if y.nnz:
        # func(fill_x, every stored y) must be the fill value, bit for bit
        probe = binary_arrays(name, scalar_t(fx), yd, a_scalar=True) if x_is_a else binary_arrays(name, yd, scalar_t(fx), b_scalar=True)
        if K.count_eq_bits(probe, fill) != y.nnz:
            return None
    if x.nnz == 0:
        return finish(x.linear_loc(), K.convert(x.data, comp_t), shape, fill, devi)
    # project x's coordinates onto y's axes (y's shape left-padded with ones; broadcast axes drop out)
    ys = (1,) * (len(shape) - y.ndim) + tuple(y.shape)
    keep = [d for d in range(len(shape)) if ys[d] != 1]
    if keep:
        ykeys = K.linearize(x.coords[keep].contiguous(), tuple(ys[d] for d in keep))
    else:
        ykeys = torch.zeros(x.nnz, dtype=torch.int64, device=devi)
    yk = y.linear_loc()      # y's own C-order keys: its size-1 axes contribute nothing, so they equal the reduced keys
    n = int(ykeys.numel())
    pos = torch.empty(n, dtype=torch.int64, device=devi)
    match = torch.empty(n + 1, dtype=torch.int64, device=devi)
    _ffi.call("spamd_lower_bound_match", n, ptr(ykeys), int(yk.numel()), ptr(yk), ptr(pos), ptr(match), stream_ptr(devi))
    yvals = _full(n, fy, comp_t, devi)
    if y.nnz:
        offs = K.exclusive_scan(match)
        cnt = int(offs[-1])
        if cnt:
            iota = torch.empty(n, dtype=torch.int64, device=devi)
            _ffi.call("spamd_iota", n, ptr(iota), stream_ptr(devi))
            hit = K.compact(iota, match, offs, cnt)                 # positions of x's elements that found a stored y
            K.scatter_into(_as_u8(yvals), hit, _as_u8(K.gather(yd, K.gather(pos, hit))))
    xd = K.convert(x.data, comp_t)
    res = binary_arrays(name, xd, yvals) if x_is_a else binary_arrays(name, yvals, xd)
    return finish(x.linear_loc(), res, shape, fill, devi)


_SAME_SHAPE_PLANS = {}   # (ufunc name, dtypes, fill bytes) -> what the fused merge needs (see `elemwise`)


def _func_name(func):
    if func is np.ndarray.astype:
        return "astype"
    return getattr(func, "__name__", None)


def _scalar_like(x):
    return np.isscalar(x) or (isinstance(x, np.ndarray) and x.ndim == 0) or (isinstance(x, torch.Tensor) and x.dim() == 0)


def _np_result(func, *args, **kwargs):
    with np.errstate(all="ignore"):
        return func(*args, **kwargs)


def _gcxs_keys2d(x):
    """64-bit keys row * C + column of a GCXS in its OWN compressed 2-D layout (cached on the array with its other derived
    layouts); None when they do not ascend strictly (`GCXS((data, indices, indptr))` takes the caller's arrays as they are)."""
    from ._dot import _validate_derived

    _validate_derived(x)
    c = x.__dict__.get("_keys2d")
    if c is None:
        if x.ndim == 1:      # (a 1-D GCXS stores its coordinates: `indices`, no pointers)
            keys = x.indices if x.indices.dtype == torch.int64 else x.indices.to(torch.int64)
        else:
            R, C = x._compressed_shape
            keys = K.csr_to_keys(x.indptr, x.indices, R, C)
        unsorted, dup = K.keys_check(keys)
        c = (None if unsorted or dup else keys,)
        x.__dict__["_keys2d"] = c
    return c[0]


def _gcxs_same_layout(name, a, b):
    """A binary ufunc on two GCXS of one shape AND one compressed layout, without leaving that layout: an elementwise
    function commutes with the axis permutation + reshape that defines the layout, so the union of the two operands' keys
    in the compressed 2-D space IS the result's storage order.  The reference converts both operands to COO, evaluates
    there and converts back (`_umath.py:40-50, 420-430`); here that cost 9 C-ABI calls and ~230 us for operands of 100
    stored elements (benchmarks/test_benchmark_coo.py:48-66; bench_small.py) against 1 call for COO operands.  Same
    stored positions, same values bit for bit (the same fused merge kernel on permuted keys).  Only for (ufunc, dtypes,
    fills) combinations the COO route has planned before (`_SAME_SHAPE_PLANS`); None = take the general route."""
    from ._convert import _pick_index_dtype
    from ._gcxs import GCXS

    if a.shape != b.shape or a.ndim < 1 or not a.size or a.compressed_axes != b.compressed_axes:
        return None
    fka, fkb = a.fill_value, b.fill_value
    plan = _SAME_SHAPE_PLANS.get((name, a.data.dtype, b.data.dtype, fka.tobytes() if hasattr(fka, "tobytes") else fka,
                                  fkb.tobytes() if hasattr(fkb, "tobytes") else fkb))
    if plan is None:
        return None
    ka, kb = _gcxs_keys2d(a), _gcxs_keys2d(b)
    if ka is None or kb is None:
        return None
    mname, comp_t, fa, fb, fill_in_kernel, fill = plan
    keys, res = merge_union(mname, ka, K.convert(a.data, comp_t), kb, K.convert(b.data, comp_t), fa, fb, fill_in_kernel)
    if a.ndim == 1:
        return GCXS((res, keys if a.indices.dtype == torch.int64 else keys.to(a.indices.dtype), ()), shape=a.shape,
                    compressed_axes=None, fill_value=fill)
    R, C = a._compressed_shape
    # (index width as the general route gives it: the first operand's, widened when the result needs it)
    indptr, indices = K.keys_to_csr(keys, R, C, _pick_index_dtype(a.indices.dtype, max(R, C, int(keys.numel()))))
    return GCXS((res, indices, indptr), shape=a.shape, compressed_axes=a.compressed_axes, fill_value=fill)


GCXS_SINGLE = True
COO_VIEW_MAX_NNZ = 1 << 18


def _coo_of(g):
    """The COO form of a GCXS operand; small operands keep it with their other derived layouts (dropped when a stored buffer
    changes): an elementwise operation on GCXS operands of different layouts or shapes converts each of them per call otherwise
    (three C-ABI calls and a read-back per operand)."""
    from ._dot import _validate_derived

    if not hasattr(g, "__dict__") or g.ndim < 2 or g.nnz > COO_VIEW_MAX_NNZ:
        return g.asformat("coo")
    _validate_derived(g)
    v = g.__dict__.get("_coo_view")
    if v is None:
        v = g.__dict__["_coo_view"] = g.asformat("coo")
    return v



def _gcxs_single(func, args, g, kwargs):
    """A function of ONE GCXS operand and scalars (`g * 2`, `abs(g)`, `g > 0.5`, `g.astype(...)`), evaluated on the operand's own
    layout: elementwise functions commute with the axis permutation + reshape that defines it, so the operand stands in as a
    COO over its compressed 2-D space (its keys there are kept with it) and the result takes the operand's `indices` /
    `indptr` as they are whenever nothing was pruned.  The reference - and rounds 1-4 here - convert to COO and back
    (`_umath.py:40-50`): 8 C-ABI calls and ~180 us for 10^3 stored elements against 3 and ~75 us for a COO operand.
    When nothing is pruned the result SHARES the operand's `indices` / `indptr` tensors (the containers never write to their
    index buffers - there is no `__setitem__` - so the sharing is only visible to code that edits `x.indices` in place)."""
    from ._coo import COO
    from ._convert import _pick_index_dtype
    from ._gcxs import GCXS

    if g.ndim < 1 or not g.size or any(not (a is g or _scalar_like(a)) for a in args):
        return None
    keys = _gcxs_keys2d(g)
    if keys is None:
        return None
    shape2 = (int(g.shape[0]),) if g.ndim == 1 else tuple(int(v) for v in g._compressed_shape)
    stand_in = COO._from_sorted_keys(keys, g.data, shape2, g.fill_value, torch.int64)
    res = elemwise(func, *[stand_in if a is g else a for a in args], **kwargs)
    if not isinstance(res, COO) or res.shape != shape2:
        return None
    if g.ndim == 1:
        rk = res.linear_loc()
        return GCXS((res.data, rk if g.indices.dtype == torch.int64 else rk.to(g.indices.dtype), ()), shape=g.shape,
                    compressed_axes=None, fill_value=res.fill_value)
    if res.nnz == g.nnz:     # nothing pruned: the same stored positions
        return GCXS((res.data, g.indices, g.indptr), shape=g.shape, compressed_axes=g.compressed_axes, fill_value=res.fill_value)
    R, C = shape2
    indptr, indices = K.keys_to_csr(res.linear_loc(), R, C, _pick_index_dtype(g.indices.dtype, max(R, C, res.nnz)))
    return GCXS((res.data, indices, indptr), shape=g.shape, compressed_axes=g.compressed_axes, fill_value=res.fill_value)


def elemwise(func, *args, **kwargs):
    """Apply `func` elementwise to sparse/dense/scalar operands (reference _umath.py:13-50)."""
    from ._coo import COO
    from ._gcxs import GCXS
    from ._sparse_array import SparseArray

    sparse_args = [a for a in args if isinstance(a, SparseArray)]
    if not sparse_args:
        raise ValueError(f"None of the args is sparse: {args}")
    out_kwargs = {}
    if all(isinstance(a, GCXS) for a in sparse_args):
        out_type = "gcxs"
        if len({a.compressed_axes for a in sparse_args}) == 1:
            out_kwargs["compressed_axes"] = sparse_args[0].compressed_axes
    else:
        out_type = "coo"
    name = _func_name(func)
    dtype_kw = kwargs.pop("dtype", None)
    if name != "astype":
        kwargs.pop("casting", None)  # values are converted explicitly; NumPy's "unsafe" semantics
    if (out_type == "gcxs" and len(args) == 2 and len(sparse_args) == 2 and dtype_kw is None and not kwargs
            and isinstance(func, np.ufunc)):     # (a plain callable that happens to be NAMED like a planned ufunc is not that ufunc)
        res = _gcxs_same_layout(name, args[0], args[1])
        if res is not None:
            return res
    if out_type == "gcxs" and len(sparse_args) == 1 and GCXS_SINGLE:
        res = _gcxs_single(func, args, sparse_args[0], dict(kwargs, **({"dtype": dtype_kw} if dtype_kw is not None else {})))
        if res is not None:
            return res
    proc = []
    for a in args:
        if isinstance(a, SparseArray):
            a = a if isinstance(a, COO) else _coo_of(a)
            if a.ndim == 0:
                a = a.todense()
        elif not (_scalar_like(a) or isinstance(a, (np.ndarray, torch.Tensor))):
            return NotImplemented
        proc.append(a)

    def finish(keys, data, shape, fill, devi):
        flags = K.flag_ne_bits(_as_u8(data), fill if data.dtype != torch.bool else np.uint8(bool(fill)))
        offs = K.exclusive_scan(flags)
        cnt = int(offs[-1])
        if cnt != data.numel():
            keys = K.compact(keys, flags, offs, cnt)
            data = K.compact(_as_u8(data), flags, offs, cnt)
            data = data.view(torch.bool) if fill.dtype == np.dtype(bool) and data.dtype == torch.uint8 else data
        ref = next(a for a in proc if isinstance(a, COO))
        out = COO._from_sorted_keys(keys, data, shape, fill, ref._index_dtype)
        return out.asformat(out_type, **out_kwargs) if out_type != "coo" else out

    coo_args = [a for a in proc if isinstance(a, COO)]
    if not coo_args:
        # every sparse operand was 0-d: the reference evaluates func on the (dense) scalars and
        # returns a 0-d array with nnz = 0 whose fill value is the result (_umath.py:435-440,481)
        vals = [a.item() if isinstance(a, torch.Tensor) else a for a in proc]
        kw = dict(kwargs)
        if dtype_kw is not None:
            kw["dtype"] = dtype_kw
        res = np.asarray(_np_result(func, *vals, **kw))
        return COO(np.empty((0, 0), dtype=np.int64), np.empty(0, dtype=res.dtype), shape=(), fill_value=res[()],
                   device=sparse_args[0].device)
    shape = coo_args[0].shape
    devi = coo_args[0].device

    # ---- unary --------------------------------------------------------------------------------
    if len(proc) == 1:
        x = proc[0]
        if name == "astype":
            target = np.dtype(dtype_kw if dtype_kw is not None else kwargs.get("dtype"))
            kwargs.pop("casting", None)
            fill = np.asarray(x.fill_value).astype(target)[()]
            return finish(x.linear_loc(), K.convert(x.data, torch_dtype(target)), shape, fill, devi)
        if func is np.invert and not kwargs and dtype_kw is None and x.data.dtype in (torch.int32, torch.int64, torch.bool):
            # ~x: the integers' bits flipped (x ^ -1), a boolean's negation - no kernel of its own, and as a function without
            # one it was evaluated on the host (69 ms at 10^7 stored elements)
            fill = _np_result(func, np.asarray(x.fill_value))[()]
            if x.data.dtype == torch.bool:
                res = unary_array("logical_not", x.data)
            else:
                res = binary_arrays("bitwise_xor", x.data.contiguous(), torch.tensor([-1], dtype=x.data.dtype, device=devi), b_scalar=True)
            return finish(x.linear_loc(), res, shape, fill, devi)
        if func is np.imag and not kwargs and dtype_kw is None and x.data.dtype in _CODE:
            # the imaginary part of a real array: no stored elements, fill 0 of the array's type
            return finish(x.linear_loc()[:0], x.data[:0], shape, _np_result(func, np.asarray(x.fill_value))[()], devi)
        if name not in _UN or kwargs or x.data.dtype not in _CODE:   # (complex / narrow value types: host-evaluated func)
            return _elemwise_general(func, proc, kwargs, dtype_kw, finish)
        fill = _np_result(func, np.asarray(x.fill_value))[()]
        data = x.data
        if data.dtype in (torch.int32, torch.int64, torch.bool) and fill.dtype.kind == "f":
            data = K.convert(data, torch_dtype(fill.dtype))
        res = unary_array(name, data)
        if dtype_kw is not None:
            res, fill = K.convert(res, torch_dtype(dtype_kw)), fill.astype(dtype_kw)
        return finish(x.linear_loc(), res, shape, np.asarray(fill)[()], devi)

    if name == "where" and len(proc) == 3 and not kwargs and dtype_kw is None and func is np.where \
            and all(isinstance(v, COO) or _scalar_like(v) for v in proc):
        try:
            return _where(proc, finish)
        except NotImplementedError:
            pass
    if len(proc) != 2 or name not in _BIN or kwargs or not isinstance(func, np.ufunc):
        # any other callable / arity / keyword: the reference's general algorithm, `func` evaluated on the host
        return _elemwise_general(func, proc, kwargs, dtype_kw, finish)
    a, b = proc
    a_sp, b_sp = isinstance(a, COO), isinstance(b, COO)
    if a_sp and b_sp and a.shape == b.shape and dtype_kw is None and a.size:
        # two canonical COO of one shape (config 1): everything the host has to decide - result dtype, compute dtype,
        # kernel op, fill values - depends only on the ufunc, the dtypes and the fills, and is decided once per combination
        # (the NumPy dtype arithmetic below costs more host time than the launch it prepares)
        fka, fkb = a.fill_value, b.fill_value
        pkey = (name, a.data.dtype, b.data.dtype, fka.tobytes() if hasattr(fka, "tobytes") else fka,
                fkb.tobytes() if hasattr(fkb, "tobytes") else fkb)
        plan = _SAME_SHAPE_PLANS.get(pkey)
        if plan is not None:
            mname, comp_t, fa, fb, fill_in_kernel, fill = plan
            keys, res = merge_union(mname, a.linear_loc(), K.convert(a.data, comp_t), b.linear_loc(), K.convert(b.data, comp_t),
                                    fa, fb, fill_in_kernel)
            out = COO._from_sorted_keys(keys, res, shape, fill, a._index_dtype)
            return out.asformat(out_type, **out_kwargs) if out_type != "coo" else out
    else:
        pkey = None

    def fill_of(v):
        if isinstance(v, COO):
            return np.asarray(v.fill_value)
        if isinstance(v, torch.Tensor):
            return np.asarray(v.item()) if v.dim() == 0 else None
        return np.asarray(v) if _scalar_like(v) else None

    def np_like(v):  # dtype-carrying stand-in for result-type computation (NEP 50 aware)
        if isinstance(v, COO):
            return np.zeros(1, dtype=v.dtype)
        if isinstance(v, torch.Tensor):
            return np.zeros(1, dtype=dev.np_dtype(v.dtype)) if v.dim() else np.asarray(v.item())
        return v if _scalar_like(v) else np.zeros(1, dtype=np.asarray(v).dtype)

    out_np = _np_result(func, np_like(a), np_like(b)).dtype
    if dtype_kw is not None:
        out_np = np.dtype(dtype_kw)
    code = _BIN[name]
    in_np = np.result_type(np_like(a), np_like(b))
    comp_np = in_np if code in _TO_BOOL_BIN or code >= 64 else out_np
    if comp_np == np.dtype(bool) and name in _BOOL_ARITH:
        # NumPy's boolean arithmetic is logical (True + True is True): never a raw uint8 add on the 0/1 bytes
        name = _BOOL_ARITH[name]
        code = _BIN[name]
    if comp_np not in (np.dtype("f4"), np.dtype("f8"), np.dtype("i4"), np.dtype("i8"), np.dtype("bool"), np.dtype("u1")) \
            or comp_np not in _COMP_TYPES.get(code, (comp_np,)) or (code in _COMP_TYPES and code < 32 and out_np != comp_np):
        return _elemwise_general(func, proc, kwargs, dtype_kw, finish)
    comp_t = torch_dtype(comp_np)

    def dev_scalar(v):
        val = v.item() if isinstance(v, (torch.Tensor, np.ndarray, np.generic)) else v
        return torch.tensor([val], dtype=comp_t, device=devi)

    # ---- sparse (x) scalar ---------------------------------------------------------------------
    if a_sp != b_sp and _scalar_like(b if a_sp else a):
        x, sc = (a, b) if a_sp else (b, a)
        scn = np.asarray(sc.item() if isinstance(sc, torch.Tensor) else sc)
        fill = _np_result(func, *( (np.asarray(x.fill_value), scn) if a_sp else (scn, np.asarray(x.fill_value)) ))
        fill = np.asarray(fill).astype(out_np)[()]
        xd = K.convert(x.data, comp_t)
        res = binary_arrays(name, xd, dev_scalar(sc), b_scalar=True) if a_sp else \
            binary_arrays(name, dev_scalar(sc), xd, a_scalar=True)
        if res.dtype != torch_dtype(out_np):
            res = K.convert(res, torch_dtype(out_np))
        return finish(x.linear_loc(), res, shape, fill, devi)

    # ---- sparse (x) dense of the same shape (the SDDMM formulation `s * (a @ b)`, A9) ---------------
    if a_sp != b_sp:
        x, d = (a, b) if a_sp else (b, a)
        dt = dev.to_device(d, devi)
        dense_shape = tuple(dt.shape)
        small_idx = None      # (a dense operand that broadcasts INTO the sparse one's shape is never materialised at that shape)
        if tuple(dt.shape) != tuple(shape):
            from ._broadcast import broadcast_shapes, broadcast_to

            common = broadcast_shapes(tuple(dt.shape), tuple(shape))
            if common != tuple(shape):   # the sparse side broadcasts too (e.g. einsum's aligned multiply)
                x = broadcast_to(x, common)
                shape = common
                dt = dt.broadcast_to(shape).contiguous()  # view + copy: memory plumbing only
            else:
                # `x * d[None, :]`, a row / column scaling (round 6): broadcasting repeats the dense values, so func(fill,
                # dense) is constant over the full shape exactly when it is over the dense operand as it stands, and a
                # stored element's partner is found from its coordinates along the axes the dense operand really has.
                # (Until then the operand was copied out at the sparse array's shape: 8 GB and 8.5 ms for a vector of 1000
                # against a 1000^3 array of 10^7 stored elements - and no memory at all for larger shapes.)
                dt = dt.contiguous()
                pad = (1,) * (len(shape) - dt.dim()) + tuple(dt.shape)
                st = (0,) * (len(shape) - dt.dim()) + tuple(dt.stride())
                small_idx = torch.zeros(int(x.nnz), dtype=torch.int64, device=devi)
                for ax in range(len(shape)):
                    if pad[ax] != 1:
                        small_idx += x.coords[ax].to(torch.int64) * int(st[ax])
        # the result stays sparse only if func(fill, dense) is constant (reference `_get_fill_value`, :505-555:
        # the fill value is element 0 of func(fills, dense); "constant" is judged with == or both-NaN, so that
        # 0 * (-1.5) = -0.0 still counts as the zero fill)
        fvx = dev_scalar(x.fill_value)
        dflat = K.convert(dt.reshape(-1), comp_t)
        allfill = binary_arrays(name, fvx, dflat, a_scalar=True) if a_sp else binary_arrays(name, dflat, fvx, b_scalar=True)
        if allfill.dtype != torch_dtype(out_np):
            allfill = K.convert(allfill, torch_dtype(out_np))
        if allfill.numel():
            fill = np.asarray(allfill[:1].cpu().numpy())[0]
            if out_np == np.dtype(bool):
                fill = np.bool_(fill)
            fill_t = allfill[:1].contiguous()
            differs = binary_arrays("not_equal", _as_u8(allfill) if out_np == np.dtype(bool) else allfill, 
                                    _as_u8(fill_t) if out_np == np.dtype(bool) else fill_t, b_scalar=True, out_bool_as=torch.uint8)
            if out_np.kind == "f" and np.isnan(fill):   # NaN == NaN for this purpose
                differs = unary_array("logical_not", unary_array("isnan", allfill)).view(torch.uint8)
            nonconst = int(differs.numel()) - K.count_eq_bits(differs, 0)
        else:
            probe_fill = _np_result(func, *((np.asarray(x.fill_value), np.zeros(1, dev.np_dtype(dt.dtype))) if a_sp
                                            else (np.zeros(1, dev.np_dtype(dt.dtype)), np.asarray(x.fill_value))))
            fill = np.asarray(probe_fill).reshape(-1)[0].astype(out_np)
            nonconst = 0
        if nonconst:
            if tuple(x.shape) != dense_shape:
                raise ValueError("Performing a mixed sparse-dense operation that would result in a dense array. "
                                 "Please make sure that func(sparse_fill_values, ndarrays) is a constant array.")
            # same shapes: the reference densifies the sparse operand and returns the dense result (:463-465)
            xdense = K.convert(x.todense_device().reshape(-1), comp_t)
            res = binary_arrays(name, xdense, dflat) if a_sp else binary_arrays(name, dflat, xdense)
            if res.dtype != torch_dtype(out_np):
                res = K.convert(res, torch_dtype(out_np))
            return res.reshape(shape)
        keys = x.linear_loc()
        dvals = K.gather(dflat, keys if small_idx is None else small_idx)
        xd = K.convert(x.data, comp_t)
        res = binary_arrays(name, xd, dvals) if a_sp else binary_arrays(name, dvals, xd)
        if res.dtype != torch_dtype(out_np):
            res = K.convert(res, torch_dtype(out_np))
        return finish(keys, res, shape, fill, devi)

    # ---- sparse (x) sparse, same shape: one sorted-key union -------------------------------------
    if a.shape != b.shape:
        from ._broadcast import broadcast_pair, broadcast_shapes

        full = broadcast_shapes(a.shape, b.shape)
        if code not in _TO_BOOL_BIN and dtype_kw is None and (tuple(a.shape) == tuple(full)) != (tuple(b.shape) == tuple(full)):
            # one operand has the broadcast shape already, the other one broadcasts into it: reduced-coordinate matching
            # (reference `_get_matching_coords`, _umath.py:310) instead of materialising its replicas
            res = _broadcast_matched(name, func, a, b, full, out_np, comp_np, finish)
            if res is not None:
                return res
        a, b = broadcast_pair(a, b)
        shape = a.shape
    fill = np.asarray(_np_result(func, np.asarray(a.fill_value), np.asarray(b.fill_value))).astype(out_np)[()]
    if a.size == 0:
        return finish(a.linear_loc(), K.convert(a.data, torch_dtype(out_np)), shape, fill, devi)
    ad, bd = K.convert(a.data, comp_t), K.convert(b.data, comp_t)
    fa, fb = np.asarray(a.fill_value).astype(comp_np), np.asarray(b.fill_value).astype(comp_np)
    if code not in _UNFUSED and (torch_dtype(out_np) == comp_t or code in _TO_BOOL_BIN):
        # default: one fused merge-path pass (function + prune inside the kernel)
        fill_in_kernel = fill if code not in _TO_BOOL_BIN else np.uint8(bool(fill))
        if pkey is not None:
            _SAME_SHAPE_PLANS[pkey] = (name, comp_t, fa, fb, fill_in_kernel, fill)
        keys, res = merge_union(name, a.linear_loc(), ad, b.linear_loc(), bd, fa, fb, fill_in_kernel)
        out = COO._from_sorted_keys(keys, res, shape, fill, a._index_dtype)
        return out.asformat(out_type, **out_kwargs) if out_type != "coo" else out
    keys, slotA, slotB = union_merge(a.linear_loc(), b.linear_loc())
    n = int(keys.numel())
    xa = _full(n, fa, comp_t, devi)
    xb = _full(n, fb, comp_t, devi)
    if a.nnz:
        K.scatter_into(_as_u8(xa), slotA, _as_u8(ad))
    if b.nnz:
        K.scatter_into(_as_u8(xb), slotB, _as_u8(bd))
    res = binary_arrays(name, xa, xb)
    if res.dtype != torch_dtype(out_np):
        res = K.convert(res, torch_dtype(out_np))
    return finish(keys, res, shape, fill, devi)
