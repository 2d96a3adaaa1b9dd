"""Raw-array kernel layer: the analogue of the reference's jitted `_dot_*` kernels.

Each function takes device tensors (raw arrays, no containers), allocates and returns fresh
outputs — the reference's ownership rule — and launches exactly one C-ABI
entry point of libsparse_amd.so (a few for the composite kernels) on the current HIP stream.  The arithmetic of the
path (products, sums, sorts, scans, merges) happens in that library; torch is used for allocation, views, dtype
casts of operands and host<->device copies.
"""
import os

import numpy as np
import torch

from . import _ffi
from ._device import code_of, np_dtype, ptr, require_hip, stream_ptr, torch_dtype


def dot_dtype(dt1, dt2):
    """Result dtype rule of every `_dot` kernel: `(zeros(dt1) * zeros(dt2)).dtype`
    (reference sparse/numba_backend/_common.py:635-636)."""
    return (np.zeros((), dtype=np_dtype(dt1)) * np.zeros((), dtype=np_dtype(dt2))).dtype


def _unify_index(*idx):
    """Index arrays handed to one kernel share a width: int32 only if all are int32."""
    want = torch.int32 if all(i.dtype == torch.int32 for i in idx) else torch.int64
    return [i if i.dtype == want else i.to(want) for i in idx], want


STREAM_MIN_ROWS = 32768   # spamd_spmm_csr's own bound for the stream form (SPAMD_ROWVEC_LDS_MIN_M, spmm_csr.hip)
STREAM_MULTI_MAX_N = 12   # widest result the stream form takes in several passes (SPAMD_STREAM_MULTI_MAX_N, sparse_amd.h)


def stream_passes(M, K, N, dtr, a_data, a_indices):
    """Passes over A the stream form of CSR x dense (csrc/spmm_stream.hip) takes for this product, 0 = another kernel has
    it.  A pass holds 4 columns (3 of 8-byte values; fewer when K x width values of B do not fit the LDS) and costs A's stream
    whatever its width - 0.16-0.18 ms at config 2's matrix, 0.25 with 8-byte values -, so (round 6, tools/r06/width_sweep.py)
    up to STREAM_MULTI_MAX_N columns are worth 3 passes (the tiled executor's padded panel: 0.77 / 1.0 ms), at most 4 columns of
    8-byte values 2 (the row-vector kernel: 0.85-0.96 ms) and at most 4 columns of 4-byte values 1 (two tie with the row-vector
    kernel).  The rule of `spamd_spmm_csr`'s own dispatch."""
    if not (1 <= N <= STREAM_MULTI_MAX_N and M >= STREAM_MIN_ROWS and K > 0):
        return 0
    passes = int(_ffi.lib().spamd_spmm_csr_stream_fits(code_of(dtr), M, K, N, ptr(a_data), ptr(a_indices)))
    worth = 3 if N > 4 else (2 if dtr.itemsize == 8 else 1)
    return passes if 1 <= passes <= worth else 0


def dot_csr_ndarray(out_shape, a_data, a_indices, a_indptr, b, *, exact=False, out=None, keep_order=False, rowvec=False):
    """C = A @ B, A in CSR, B dense row-major, C dense — `_dot_csr_ndarray`
    (reference _common.py:720-755).  `exact=True` reproduces the reference's separate
    multiply/add bit-for-bit; the default uses one FMA per term (same k-ascending order; results of at most 4 columns
    are summed in a per-row tree order by the stream / row-vector kernels unless `keep_order`; `rowvec` keeps the
    row-vector kernel where the stream form would run)."""
    M, N = int(out_shape[0]), int(out_shape[1])
    dev = require_hip(a_data, a_indices, a_indptr, b)
    dtr = torch_dtype(dot_dtype(a_data.dtype, b.dtype))
    vcode = code_of(dtr)
    if a_data.dtype != dtr:
        a_data = a_data.to(dtr)
    if b.dtype != dtr:
        b = b.to(dtr)
    b = b.contiguous()  # the reference's np.ascontiguousarray(b), _common.py:744
    if b.dim() != 2 or b.shape[1] != N:
        raise ValueError(f"dense operand has shape {tuple(b.shape)}, expected (K, {N})")
    K = int(b.shape[0])
    (a_indices, a_indptr), it = _unify_index(a_indices.contiguous(), a_indptr.contiguous())
    if a_indptr.numel() != M + 1:
        raise ValueError(f"indptr has {a_indptr.numel()} entries, expected {M + 1}")
    a_data = a_data.contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=dtr, device=dev)
    elif out.shape != (M, N) or out.dtype != dtr or not out.is_contiguous():
        raise ValueError("out buffer has wrong shape/dtype/layout")
    if (not keep_order and not rowvec and not (exact and dtr.is_floating_point)
            and stream_passes(M, K, N, dtr, a_data, a_indices)):
        # what spamd_spmm_csr's own dispatch would pick, with the number of stored elements handed over (the kernel need
        # not read it from indptr at the head of every wave's start-up chain)
        _ffi.call("spamd_spmm_csr_stream", vcode, code_of(it), M, K, N, ptr(a_data), ptr(a_indices), ptr(a_indptr),
                  ptr(b), N, ptr(out), N, int(a_data.numel()), 0, stream_ptr(dev))
        return out
    _ffi.call("spamd_spmm_csr", vcode, code_of(it), M, K, N, ptr(a_data), ptr(a_indices),
              ptr(a_indptr), ptr(b), max(N, 1), ptr(out), max(N, 1),
              (_ffi.EXACT_MULADD if exact else 0) | (_ffi.SPMM_ROWGROUP if keep_order else 0) |
              (_ffi.SPMM_ROWVEC if rowvec else 0), stream_ptr(dev))
    return out


def transposed_copy(x):
    """`x.t().contiguous()` of a dense 2-D tensor through csrc/transpose.hip (64 x 64 LDS tiles: torch's strided copy runs
    at 1.3 TB/s on a 128 x 10^6 operand - a third of the `dense @ sparse` product it prepares, round 6); other layouts and
    element sizes keep torch's copy."""
    if (not isinstance(x, torch.Tensor) or x.dim() != 2 or not x.is_cuda or not x.is_contiguous() or x.numel() == 0
            or x.element_size() not in (1, 2, 4, 8) or x.is_complex()):
        return x.t().contiguous()
    out = torch.empty((x.shape[1], x.shape[0]), dtype=x.dtype, device=x.device)
    _ffi.call("spamd_transpose_2d", x.element_size(), int(x.shape[0]), int(x.shape[1]), ptr(x), int(x.shape[1]), ptr(out),
              int(x.shape[0]), stream_ptr(x.device))
    return out


def has_nan(data):
    """Any NaN in a float tensor?  (reference `nan_check`, _common.py:51-69).  One streaming
    pass on the device; the 4-byte flag is the only thing copied back."""
    dev = require_hip(data)
    if data.numel() == 0 or not data.is_floating_point():
        return False
    data = data.contiguous()
    if data.data_ptr() % 16:
        data = data.clone()
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    _ffi.call("spamd_has_nan", code_of(data.dtype), data.numel(), ptr(data), ptr(flag), stream_ptr(dev))
    return bool(flag.item())


class _WordMailbox:
    """Per device: pinned host slots the device delivers a few int64 words into (C ABI `spamd_deliver_words`), so that a
    data-dependent size reaches the host without `tolist()`'s blocking copy (a stream synchronisation plus a copy command,
    ~20 us).  Slots are used round-robin with a per-call sequence number as the marker: the late store of an abandoned
    call cannot be mistaken for the current one's."""

    _pool = {}
    SLOTS, WIDTH = 64, 17

    def __init__(self):
        import threading

        self.pinned = torch.zeros(self.SLOTS * self.WIDTH, dtype=torch.int64).pin_memory()
        self.view = self.pinned.numpy()
        self.seq = 0
        self.lock = threading.Lock()     # slots are handed out under it (two host threads must not share one)

    def take(self):
        with self.lock:
            self.seq += 1
            return self.seq % self.SLOTS, self.seq

    @classmethod
    def get(cls, dev):
        m = cls._pool.get(dev.index)
        if m is None:
            m = cls._pool[dev.index] = cls()
        return m


def read_words(t):
    """The int64 words of a small device tensor (<= 16) as Python ints, behind everything queued on the current stream."""
    n = int(t.numel())
    dev = t.device
    if n > 16 or t.dtype != torch.int64 or not t.is_contiguous():
        return [int(v) for v in t.tolist()]
    m = _WordMailbox.get(dev)
    k, marker = m.take()
    base = k * m.WIDTH
    _ffi.call("spamd_deliver_words", ptr(t), n, m.pinned.data_ptr() + 8 * base, marker, stream_ptr(dev))
    view = m.view
    for _ in range(4_000_000):        # (~0.1 s; then a synchronisation decides)
        if int(view[base + n]) == marker:
            break
    else:
        torch.cuda.current_stream(dev).synchronize()
        if int(view[base + n]) != marker:
            raise _ffi.HipBackendError("spamd_deliver_words did not deliver")
    return [int(v) for v in view[base: base + n]]


class NanProbe:
    """A NaN scan in flight: the kernel writes its verdict into a pinned host int; `result()` waits for the event
    recorded right behind the scan — NOT for whatever was queued after it — and reads the int."""

    # verdict slots: ONE pinned int32 slab, handed out by index (a pinned allocation per probe costs ~0.1 ms of host time,
    # which a loop that runs many products ahead of the device would pay once per product in flight)
    _SLOTS = 1024
    _slab = None      # pinned int32[_SLOTS]
    _view = None      # its NumPy view: host-side reads / writes of a verdict without a tensor op
    _free = []        # free slot indices
    _events = []      # reusable events

    @classmethod
    def _take(cls):
        if cls._slab is None:
            cls._slab = torch.zeros(cls._SLOTS, dtype=torch.int32).pin_memory()
            cls._view = cls._slab.numpy()
            cls._free = list(range(cls._SLOTS))
        if not cls._free:
            return None
        return cls._free.pop()

    def __init__(self, data):
        dev = require_hip(data)
        self.slot = NanProbe._take()
        if self.slot is None:            # more than _SLOTS scans in flight: a private pinned word
            self.own = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.view, self.at, addr = self.own.numpy(), 0, self.own.data_ptr()
        else:
            self.own = None
            self.view, self.at, addr = NanProbe._view, self.slot, NanProbe._slab.data_ptr() + 4 * self.slot
        self.event = NanProbe._events.pop() if NanProbe._events else torch.cuda.Event()
        self.view[self.at] = 0
        self.done = None
        self.launched = False
        self._keep = data  # the scanned buffer must outlive the kernel
        try:
            _ffi.call("spamd_has_nan_async", code_of(data.dtype), data.numel(), ptr(data), addr, stream_ptr(dev))
            self.launched = True
            # behind the scan on ITS stream: the array may live on another device than the current one (`to_device`)
            self.event.record(torch.cuda.current_stream(dev))
            self.recorded = True
        except BaseException:
            self.discard()
            raise

    def discard(self):
        """Give the verdict slot and the event back without reading the verdict (the launch failed, or the product
        this scan belonged to raised before asking)."""
        if self.done is None:
            self.done = False
            if self.launched:
                # the scan may still be running and will write its verdict word: the slot (and the scanned buffer) go back
                # only once it has finished - behind the event when that was recorded, else behind the whole device
                if getattr(self, "recorded", False):
                    self.event.synchronize()
                else:
                    torch.cuda.synchronize()
            if self.slot is not None:
                NanProbe._free.append(self.slot)
            if self.event is not None:
                NanProbe._events.append(self.event)
            self._keep = self.view = self.event = self.own = None

    def ready(self):
        """True once the scan has finished (never blocks)"""
        return self.done is not None or self.event.query()

    def result(self):
        if self.done is None:
            self.event.synchronize()
            self.done = bool(int(self.view[self.at]))
            if self.slot is not None:
                NanProbe._free.append(self.slot)
            NanProbe._events.append(self.event)
            self._keep = self.view = self.event = self.own = None
        return self.done


def has_nan_async(data):
    """`has_nan` without draining the stream: returns a NanProbe (or False when `data` cannot hold a NaN)."""
    if data.numel() == 0 or not data.is_floating_point():
        return False
    data = data.contiguous()
    if data.data_ptr() % 16:
        data = data.clone()
    return NanProbe(data)


# ---------------------------------------------------------------------------------------------
# key primitives (prims.hip): thin wrappers, one C-ABI call each
# ---------------------------------------------------------------------------------------------
import ctypes as _ct


def _harr64(vals):
    return (_ct.c_int64 * max(len(vals), 1))(*[int(v) for v in vals])


def _harr32(vals):
    return (_ct.c_int32 * max(len(vals), 1))(*[int(v) for v in vals])


def c_strides(shape):
    """C-order element strides of `shape` (host ints)."""
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= int(d)
    return list(reversed(st))


def _check_ndim(n):
    if n > _ffi.MAX_NDIM:
        raise NotImplementedError(f"arrays with more than {_ffi.MAX_NDIM} dimensions are not supported by the hip backend")


def _key_bits(max_key):
    return max(1, int(max_key).bit_length())


def index_dtype_ok(t):
    return t.dtype in (torch.int32, torch.int64)


def linearize(coords, shape, axis_order=None):
    """C-order keys of `coords[axis_order]` w.r.t. `shape[axis_order]` — `linear_loc`
    (reference _coo/common.py:56-64)."""
    dev = require_hip(coords)
    ndim, nnz = int(coords.shape[0]), int(coords.shape[1])
    _check_ndim(ndim)
    order = list(range(ndim)) if axis_order is None else [int(a) for a in axis_order]
    rshape = [int(shape[a]) for a in order]
    keys = torch.empty(nnz, dtype=torch.int64, device=dev)
    if ndim == 0 or nnz == 0:
        return keys.zero_()
    coords = coords.contiguous()
    _ffi.call("spamd_coo_linearize", code_of(coords.dtype), ndim, nnz, ptr(coords), nnz,
              _harr64(c_strides(rshape)), _harr32(order), ptr(keys), stream_ptr(dev))
    return keys


def delinearize(keys, shape, idx_dtype=torch.int64):
    """keys -> coords[ndim, nnz] for `shape` (the div/mod chain of reference core.py:1090-1098)."""
    dev = require_hip(keys)
    ndim, nnz = len(shape), int(keys.numel())
    _check_ndim(ndim)
    coords = torch.empty((ndim, nnz), dtype=idx_dtype, device=dev)
    if ndim and nnz:
        _ffi.call("spamd_coo_delinearize", code_of(idx_dtype), ndim, nnz, ptr(keys), _harr64(c_strides(shape)),
                  _harr64(shape), ptr(coords), nnz, stream_ptr(dev))
    return coords


def permute_keys(keys, src_shape, perm):
    """Keys of the array transposed with `axes=perm` (destination axis d <- source axis perm[d])."""
    dev = require_hip(keys)
    ndim = len(src_shape)
    _check_ndim(ndim)
    if list(perm) == list(range(ndim)) or keys.numel() == 0:
        return keys
    out = torch.empty_like(keys)
    _ffi.call("spamd_permute_keys", ndim, keys.numel(), ptr(keys), _harr64(c_strides(src_shape)),
              _harr64(src_shape), _harr32(perm), ptr(out), stream_ptr(dev))
    return out


LEAD_LAST = True               # reductions over the leading axes: the kept-axes-first order by merging the slabs (csrc/lead_rotate.hip)
LEAD_LAST_RANGE = 2048         # elements a cell range should hold (the kernel's arrays take 4096)
LEAD_LAST_MAX_NNZ = 1 << 22     # beyond this the radix sort wins (it streams; the merge is a latency chain per range): measured at
#                                S = 1000, P = 10^6, whole reduction by merge / by sort: 10^6 elements 0.15 / 0.22 ms, 2 x 10^6: 0.22 / 0.25,
#                                10^7: merge kernel alone 0.43 ms against 0.64 ms for everything with the sort
LEAD_LAST_MIN_SLABS = 64
LEAD_LAST_MAX_BOUNDS = 1 << 20  # boundary words (runs x (ranges + 1)) of the slab merge: 4 MB
LEAD_LAST_STATS = {}


def lead_last_plan(n, n_slabs, n_cells, max_slabs=2048, max_cells=2048):
    """(cells per range, ranges) of the slab merge for n elements in n_slabs sorted runs over n_cells kept cells, or None when
    the problem is the sort's: too many runs, or a key space so much larger than the element count that most workgroups (one
    per range) and most boundary words (one per run and range) would be empty.  A range aims at LEAD_LAST_RANGE elements."""
    if n <= 0 or n > LEAD_LAST_MAX_NNZ or n_slabs > max_slabs or n_cells >= 2 ** 42 or n_cells < 1:
        return None
    # (a run's piece of a range is walked by ONE thread: a range holds at most ~32 elements per run, and below 64 runs a
    # workgroup's 512 threads have nothing to share - the sort)
    if n_slabs < LEAD_LAST_MIN_SLABS:
        return None
    target = min(LEAD_LAST_RANGE, 32 * n_slabs)
    cells = 1
    while cells * 2 <= max_cells and cells * 2 * n <= target * n_cells:
        cells *= 2
    ranges = -(-n_cells // cells)
    # (every range is a workgroup, every (run, range) a boundary word found by a binary search of the keys)
    if ranges > max(4096, n // 64) or n_slabs * (ranges + 1) > LEAD_LAST_MAX_BOUNDS:
        return None
    return cells, ranges


def keys_lead_last(keys, vals, n_slabs, n_cells, failed):
    """(keys', vals') of a canonical COO whose leading axes (n_slabs index values) are moved behind the kept ones (n_cells):
    keys' = cell * n_slabs + slab, ascending - what `permute_keys` + `sort_key_value` return, without the sort.  None when
    the shape of the problem is not the kernel's (`lead_last_plan`); `failed`: a device int64 word the kernels set when a range
    was too full (the caller reads it with its own read-back and takes the sort then)."""
    dev = require_hip(keys, vals)
    n = int(keys.numel())
    lim = _ffi.lib().spamd_keys_lead_last_limits
    plan = lead_last_plan(n, n_slabs, n_cells, int(lim(0)), int(lim(1))) if LEAD_LAST and vals.element_size() in (4, 8) else None
    if plan is None:
        return None
    cells, ranges = plan
    bounds = torch.empty(n_slabs * (ranges + 1), dtype=torch.int32, device=dev)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    _ffi.call("spamd_keys_lead_last", vals.element_size(), n, ptr(keys.contiguous()), ptr(vals.contiguous()), int(n_slabs), int(n_cells),
              cells, ptr(bounds), ptr(ko), ptr(vo), ptr(failed), stream_ptr(dev))
    LEAD_LAST_STATS.update(cells=cells, ranges=ranges, calls=LEAD_LAST_STATS.get("calls", 0) + 1)
    return ko, vo


def keys_check(keys):
    """(not_sorted, has_duplicates) of a key array — the `np.diff(linear)` checks of reference
    core.py:1310-1313,1331-1338.  Synchronises (8 bytes copied back)."""
    dev = require_hip(keys)
    if keys.numel() < 2:
        return False, False
    flags = torch.empty(2, dtype=torch.int32, device=dev)
    _ffi.call("spamd_keys_check", keys.numel(), ptr(keys), ptr(flags), stream_ptr(dev))
    f = flags.tolist()
    return bool(f[0]), bool(f[1])


def coords_in_range(coords, shape):
    """True if every coordinate lies in [0, shape[d]) — one read-only pass, 4 bytes copied back."""
    dev = require_hip(coords)
    ndim, nnz = int(coords.shape[0]), int(coords.shape[1])
    if ndim == 0 or nnz == 0:
        return True
    _check_ndim(ndim)
    coords = coords.contiguous()
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    _ffi.call("spamd_coords_check", code_of(coords.dtype), ndim, nnz, ptr(coords), nnz, _harr64(shape), ptr(flag),
              stream_ptr(dev))
    return int(flag.item()) == 0


def sort_keys(keys, max_key):
    """Stable sort of int64 keys in [0, max_key]; returns (sorted_keys, perm) with
    sorted_keys == keys[perm] — `np.argsort(kind="mergesort")` (reference core.py:1315)."""
    dev = require_hip(keys)
    n = int(keys.numel())
    if n == 0:
        return keys, torch.empty(0, dtype=torch.int64, device=dev)
    iota = torch.empty(n, dtype=torch.int64, device=dev)
    _ffi.call("spamd_iota", n, ptr(iota), stream_ptr(dev))
    ws_bytes = int(_ffi.lib().spamd_sort_pairs_ws_bytes(n))
    if ws_bytes < 0:
        raise _ffi.HipBackendError(f"spamd_sort_pairs_ws_bytes failed: {ws_bytes}")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    ko, po = torch.empty_like(keys), torch.empty_like(iota)
    _ffi.call("spamd_sort_pairs", n, ptr(keys.contiguous()), ptr(ko), ptr(iota), ptr(po), _key_bits(max_key),
              ptr(ws), ws_bytes, stream_ptr(dev))
    return ko, po


def sort_key_value(keys, vals, max_key):
    """Stable sort of (int64 key, value) pairs by key; values are moved bit-wise."""
    dev = require_hip(keys, vals)
    n = int(keys.numel())
    if n == 0:
        return keys, vals
    ws_bytes = int(_ffi.lib().spamd_sort_kv_ws_bytes(vals.element_size(), n))
    if ws_bytes < 0:
        raise _ffi.HipBackendError(f"spamd_sort_kv_ws_bytes failed: {ws_bytes}")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    _ffi.call("spamd_sort_kv", vals.element_size(), n, ptr(keys.contiguous()), ptr(ko), ptr(vals.contiguous()), ptr(vo),
              _key_bits(max_key), ptr(ws), ws_bytes, stream_ptr(dev))
    return ko, vo


def exclusive_scan(flags):
    """`flags` holds n+1 int64 entries (the last is ignored); returns offsets[n+1] with
    offsets[n] = total."""
    dev = require_hip(flags)
    n = int(flags.numel()) - 1
    out = torch.empty_like(flags)
    ws_bytes = int(_ffi.lib().spamd_scan_ws_bytes(n))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _ffi.call("spamd_exclusive_scan", n, ptr(flags), ptr(out), ptr(ws), ws_bytes, stream_ptr(dev))
    return out


def new_flags(n, dev):
    """int64[n+1] flag buffer for flag_* -> exclusive_scan -> compact."""
    return torch.empty(n + 1, dtype=torch.int64, device=dev)


def flag_heads(keys):
    dev = require_hip(keys)
    f = new_flags(keys.numel(), dev)
    _ffi.call("spamd_flag_heads", keys.numel(), ptr(keys), ptr(f), stream_ptr(dev))
    return f


def flag_ne_bits(data, fill_value):
    """flags[i] = data[i] is not bit-identical to fill_value (reference `equivalent` without
    `loose`, _utils.py:448-452)."""
    dev = require_hip(data)
    f = new_flags(data.numel(), dev)
    lo, hi = _fill_words(fill_value, np_dtype(data.dtype))
    _ffi.call("spamd_flag_ne_bits", data.element_size(), data.numel(), ptr(data.contiguous()), lo, hi, ptr(f),
              stream_ptr(dev))
    return f


DENSE_NONFILL = True
DENSE_NONFILL_DIRECT = 1 << 24     # elements up to which the outputs are allocated for the worst case (12-16 bytes each) and trimmed


def dense_nonfill(flat, fill_value, numeric=False):
    """(keys, values) of the elements of a flat dense tensor that are not bit-identical to `fill_value`, keys ascending - one
    pass (csrc/prims.hip `spamd_dense_nonfill`) instead of flags + scan + iota + two compactions.  None for element sizes the
    kernel does not take (complex128) and for inputs so large that worst-case outputs would not be reasonable."""
    dev = require_hip(flat)
    n = int(flat.numel())
    if not DENSE_NONFILL or flat.element_size() not in (1, 2, 4, 8) or n == 0 or n > DENSE_NONFILL_DIRECT:
        return None
    lo, _ = _fill_words(fill_value, np_dtype(flat.dtype))
    mask = (1 << 64) - 1
    if numeric:     # `value != 0`: +0 and -0 are one value (NaN is not zero: kept, as NumPy's comparison keeps it)
        if lo != 0:
            return None
        if flat.dtype.is_floating_point:
            mask = (1 << (8 * flat.element_size() - 1)) - 1
    work = torch.empty(int(_ffi.lib().spamd_dense_nonfill_work_words(n)), dtype=torch.int64, device=dev)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    vals = torch.empty(n, dtype=flat.dtype, device=dev)
    _ffi.call("spamd_dense_nonfill", flat.element_size(), n, ptr(flat.contiguous()), lo, mask, ptr(work), ptr(keys), ptr(vals), stream_ptr(dev))
    count = int(work[1])
    if count * 4 < n * 3:
        return keys[:count].clone(), vals[:count].clone()
    return keys[:count], vals[:count]


def _fill_words(fill_value, npdt):
    """The fill value's bit pattern as (low 8 bytes, bytes 8..15): the second word is only non-zero for 16-byte
    elements (complex128: real part, imaginary part)."""
    raw = np.asarray(fill_value, dtype=npdt).reshape(1)
    if npdt.itemsize == 16:
        w = raw.view(np.uint64)
        return int(w[0]), int(w[1])
    return int(raw.view(f"u{npdt.itemsize}")[0]), 0


def note_zero_bits_count(data, count):
    """Remember (on the tensor, for as long as it is not written to) how many of its elements have all-zero bits — a
    producer that has just written `data` knows it for free (the SpGEMM pack kernel), and `count_eq_bits(data, 0)`, i.e.
    the prune of a zero-fill container built from it, then costs one device word instead of a pass over the data."""
    data._zero_bits_count = (count, data._version)


def known_eq_bits(data, fill_value):
    """The number of stored elements bit-identical to `fill_value` when the producer of `data` left it on the tensor
    (`note_zero_bits_count`: all-zero bits only) and the tensor was not written since; None otherwise.  No device work."""
    known = getattr(data, "_zero_bits_count", None)
    if known is None or known[1] != data._version or data.numel() == 0:
        return None
    bits, bits_hi = _fill_words(fill_value, np_dtype(data.dtype))
    return known[0] if bits == 0 and bits_hi == 0 else None


def count_eq_bits(data, fill_value):
    """Number of stored elements bit-identical to fill_value (one read-only pass)."""
    dev = require_hip(data)
    if data.numel() == 0:
        return 0
    bits, bits_hi = _fill_words(fill_value, np_dtype(data.dtype))
    known = getattr(data, "_zero_bits_count", None)
    if bits == 0 and bits_hi == 0 and known is not None and known[1] == data._version:
        return int(known[0])
    c = torch.empty(1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_count_eq_bits", data.element_size(), data.numel(), ptr(data.contiguous()), bits, bits_hi, ptr(c),
              stream_ptr(dev))
    return int(c[0])


PRUNE_COUNT_FIRST = 1 << 22   # above this many elements a prune first counts the fill values (usually zero)


def compact(src, flags, offsets, count):
    """Stream compaction of a 1-D tensor or of every row of a 2-D [k, n] tensor."""
    dev = require_hip(src)
    n = int(src.shape[-1])
    if src.dim() == 1:
        dst = torch.empty(count, dtype=src.dtype, device=dev)
        _ffi.call("spamd_compact", src.element_size(), n, ptr(src.contiguous()), ptr(flags), ptr(offsets), ptr(dst),
                  stream_ptr(dev))
        return dst
    src = src.contiguous()
    dst = torch.empty((src.shape[0], count), dtype=src.dtype, device=dev)
    _ffi.call("spamd_compact_rows", src.element_size(), int(src.shape[0]), n, ptr(src), n, ptr(flags), ptr(offsets), ptr(dst),
              count, stream_ptr(dev))   # all rows of the [k, n] matrix in one launch
    return dst


def gather(src, perm):
    """src[..., perm] for 1-D or [k, n] tensors."""
    dev = require_hip(src, perm)
    n = int(perm.numel())
    if src.dim() == 1:
        dst = torch.empty(n, dtype=src.dtype, device=dev)
        _ffi.call("spamd_gather", src.element_size(), n, ptr(src.contiguous()), ptr(perm), ptr(dst), stream_ptr(dev))
        return dst
    src = src.contiguous()
    dst = torch.empty((src.shape[0], n), dtype=src.dtype, device=dev)
    _ffi.call("spamd_gather_rows", src.element_size(), int(src.shape[0]), n, ptr(src), int(src.shape[1]), ptr(perm), ptr(dst), n,
              stream_ptr(dev))
    return dst


def scatter_into(dst_flat, keys, src):
    dev = require_hip(dst_flat, keys, src)
    _ffi.call("spamd_scatter", src.element_size(), keys.numel(), ptr(src.contiguous()), ptr(keys), ptr(dst_flat),
              stream_ptr(dev))
    return dst_flat


def keys_to_csr(keys, R, C, idx_dtype):
    dev = require_hip(keys)
    nnz = int(keys.numel())
    indptr = torch.empty(R + 1, dtype=idx_dtype, device=dev)
    indices = torch.empty(nnz, dtype=idx_dtype, device=dev)
    _ffi.call("spamd_keys_to_csr", code_of(idx_dtype), nnz, ptr(keys), R, C, ptr(indptr), ptr(indices), stream_ptr(dev))
    return indptr, indices


def csr_to_keys(indptr, indices, R, C):
    dev = require_hip(indptr, indices)
    (indptr, indices), it = _unify_index(indptr.contiguous(), indices.contiguous())
    nnz = int(indices.numel())
    keys = torch.empty(nnz, dtype=torch.int64, device=dev)
    _ffi.call("spamd_csr_to_keys", code_of(it), R, nnz, ptr(indptr), ptr(indices), C, ptr(keys), stream_ptr(dev))
    return keys


def rows_to_indptr(rows, R):
    """int64 indptr of sorted row ids (`cumsum(bincount(coords[0]))`, reference _common.py:452-458)."""
    dev = require_hip(rows)
    if not index_dtype_ok(rows):
        rows = rows.to(torch.int64)
    indptr = torch.empty(R + 1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_rows_to_indptr", code_of(rows.dtype), rows.numel(), ptr(rows.contiguous()), R, ptr(indptr),
              stream_ptr(dev))
    return indptr


_CODE_T = {torch.float32: _ffi.F32, torch.float64: _ffi.F64, torch.int32: _ffi.I32, torch.int64: _ffi.I64,
           torch.bool: _ffi.U8, torch.uint8: _ffi.U8}


def convert(t, dtype):
    """astype on the device between float32/float64/int32/int64/bool."""
    dtype = torch_dtype(dtype)
    if t.dtype == dtype:
        return t
    dev = require_hip(t)
    if t.dtype not in _CODE_T or dtype not in _CODE_T:
        raise TypeError(f"hip backend cannot convert {t.dtype} -> {dtype}")
    out = torch.empty(t.shape, dtype=dtype, device=dev)
    _ffi.call("spamd_convert", _CODE_T[t.dtype], _CODE_T[dtype], t.numel(), ptr(t.contiguous()), ptr(out),
              stream_ptr(dev))
    return out


# ---------------------------------------------------------------------------------------------
# the rest of the `_dot_*` kernel family (reference _common.py:758-1158), built on the kernels above
# ---------------------------------------------------------------------------------------------
def csx_swap_2d(data, indices, indptr, n_major, n_minor):
    """Re-compress a 2-D compressed matrix along its other axis (CSR <-> CSC): returns (data, indices, indptr) with
    n_minor + 1 pointers.  The input is ordered by (major, minor), so a STABLE sort on the minor index alone gives
    (minor, major) order: ceil(log2(n_minor)) key bits instead of log2(n_major * n_minor), and the major ids and
    4-byte values ride along packed into one 8-byte payload (no permutation + gathers).
    (`_transpose` / change_compressed_axes, reference _compressed/convert.py:210-273, compressed.py:388-423)."""
    dev = require_hip(data, indices, indptr)
    nnz = int(indices.numel())
    it = indices.dtype if index_dtype_ok(indices) else torch.int64
    if nnz == 0:
        return data, indices.to(it), torch.zeros(n_minor + 1, dtype=it, device=dev)
    if data.element_size() in (4, 8) and n_major < 2 ** 31 and n_minor < 2 ** 31 and index_dtype_ok(indices) \
            and indices.dtype == indptr.dtype and not data.is_complex():
        # one library call: pack (32-bit minor key, major id | value bits), stable sort on the key, unpack + pointers
        # (8-byte values - the reference's default float64 - ride in a 16-byte payload: round 4; before, they went through a
        # 64-bit key sort with a permutation payload and two gathers: ~8 ms per 10^8 elements)
        wide = data.element_size() == 8
        ws_bytes = int((_ffi.lib().spamd_csx_swap8_ws_bytes if wide else _ffi.lib().spamd_csx_swap_ws_bytes)(nnz))
        if ws_bytes < 0:
            raise _ffi.HipBackendError(f"spamd_csx_swap_ws_bytes failed: {ws_bytes}")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        new_data, new_indices = torch.empty_like(data.contiguous()), torch.empty(nnz, dtype=it, device=dev)
        new_indptr = torch.empty(n_minor + 1, dtype=it, device=dev)
        _ffi.call("spamd_csx_swap8" if wide else "spamd_csx_swap", code_of(it), int(n_major), int(n_minor), nnz, ptr(data.contiguous()), ptr(indices.contiguous()),
                  ptr(indptr.contiguous()), ptr(new_data), ptr(new_indices), ptr(new_indptr), ptr(ws), ws_bytes, stream_ptr(dev))
        return new_data, new_indices, new_indptr
    major = csr_to_keys(indptr, torch.zeros_like(indices), n_major, 1)     # major id of every stored element
    keys = convert(indices.contiguous(), torch.int64)
    if data.element_size() == 4 and n_major < 2 ** 31:
        payload = torch.stack([major.to(torch.int32), data.contiguous().view(torch.int32)], dim=1).view(torch.int64)
        keys, payload = sort_key_value(keys, payload.reshape(-1), max(n_minor - 1, 1))
        pr = payload.view(torch.int32).reshape(nnz, 2)
        new_indices = pr[:, 0].contiguous().to(it)
        new_data = pr[:, 1].contiguous().view(data.dtype)
    else:
        keys, perm = sort_keys(keys, max(n_minor - 1, 1))
        new_indices = gather(major, perm).to(it)
        new_data = gather(data, perm)
    new_indptr = rows_to_indptr(keys, n_minor).to(it)
    return new_data, new_indices, new_indptr


def _csc_to_csr(a_shape, a_data, a_indices, a_indptr):
    """(M x K) stored by columns -> CSR arrays.  Stable: within a row the columns ascend, so
    the CSR product accumulates in the same k order as the reference's column sweep."""
    M, Kd = int(a_shape[0]), int(a_shape[1])
    return csx_swap_2d(a_data, a_indices, a_indptr, Kd, M)


def dot_csc_ndarray(a_shape, b_shape, a_data, a_indices, a_indptr, b, *, exact=False):
    """C = A @ B with A stored by columns — `_dot_csc_ndarray` (reference _common.py:869-904).
    The reference scatters `out[ind, :] += v * b[i, :]` column by column; here A is re-compressed
    by rows on the device (a stable key sort) and the CSR kernel is used: same per-element
    summation order, no atomics, deterministic."""
    data, indices, indptr = _csc_to_csr(a_shape, a_data, a_indices, a_indptr)
    return dot_csr_ndarray((int(a_shape[0]), int(b_shape[1])), data, indices, indptr, b, exact=exact)


def dot_coo_ndarray(coords, data, b, out_shape, *, exact=False):
    """C = S @ B, S a row-sorted 2-D COO — `_dot_coo_ndarray` (reference _common.py:979-1014).
    Takes B itself (K x N); the reference kernel receives the transposed view."""
    M = int(out_shape[0])
    indptr = rows_to_indptr(coords[0], M)
    cols = coords[1]
    return dot_csr_ndarray(out_shape, data, cols, indptr, b, exact=exact)


def coo_transposed_csr(coords, data, Kd, N):
    """S^T compressed by rows - (data, indices, indptr) - of a 2-D COO S (Kd x N): what `dot_ndarray_coo` multiplies with.
    Depends on S alone; `_dot` keeps it with the operand between products."""
    keys = linearize(coords, (Kd, N), axis_order=(1, 0))    # col * K + row
    keys, perm = sort_keys(keys, max(Kd * N - 1, 1))
    it = coords.dtype if index_dtype_ok(coords) else torch.int64
    indptr, indices = keys_to_csr(keys, N, Kd, it)
    return gather(data, perm), indices, indptr


def dot_ndarray_coo(a, coords, data, out_shape, *, exact=False, st=None):
    """C = A @ S, A dense (M x K), S 2-D COO (K x N) — `_dot_ndarray_coo`
    (reference _common.py:1075-1103), computed as (S^T @ A^T)^T with S^T compressed by rows (`st`: that form, when the
    caller has it already)."""
    M, N = int(out_shape[0]), int(out_shape[1])
    Kd = int(a.shape[1])
    tdata, indices, indptr = st if st is not None else coo_transposed_csr(coords, data, Kd, N)
    res = dot_csr_ndarray((N, M), tdata, indices, indptr, a.t().contiguous(), exact=exact)
    return res.t()


def _sparsify(dense, struct_mask=None, numeric=False):
    """Dense 2-D tensor -> (keys, data) of the entries to store.  `struct_mask` (same shape, any
    dtype, nonzero = structurally present) restricts them; `numeric=True` keeps value != 0
    (the COO variants' `if data_curr != 0`), otherwise bit-pattern != +0 (GCXS prune)."""
    flat = dense.reshape(-1).contiguous()
    if struct_mask is None and flat.dtype in (torch.float32, torch.float64, torch.int32, torch.int64):
        fused = dense_nonfill(flat, 0, numeric=numeric)
        if fused is not None:
            return fused
    if numeric:
        from ._umath import binary_arrays

        nz = binary_arrays("not_equal", flat, torch.zeros(1, dtype=flat.dtype, device=flat.device), b_scalar=True)
        flags = flag_ne_bits(nz.view(torch.uint8), 0)
    else:
        flags = flag_ne_bits(flat, 0)
    if struct_mask is not None:
        from ._umath import binary_arrays

        sm = flag_ne_bits(struct_mask.reshape(-1).contiguous(), 0)
        flags = binary_arrays("multiply", flags, sm)
    offs = exclusive_scan(flags)
    cnt = int(offs[-1])
    n = flat.numel()
    iota = torch.empty(n, dtype=torch.int64, device=flat.device)
    _ffi.call("spamd_iota", n, ptr(iota), stream_ptr(flat.device))
    return compact(iota, flags, offs, cnt), compact(flat, flags, offs, cnt)


def _pattern_product(out_shape, a_indices, a_indptr, b, csc_shape=None):
    """Structural non-zero test of the reference's sparse-returning variants: entry (i, j) is
    stored iff some b[k, j] != 0 with k in row i of A (_common.py:796-798, 843-844)."""
    from ._umath import binary_arrays

    bz = binary_arrays("not_equal", b.reshape(-1).contiguous(), torch.zeros(1, dtype=b.dtype, device=b.device),
                       b_scalar=True)
    bz = convert(bz, torch.float32).reshape(b.shape)
    ones = torch.ones(a_indices.numel(), dtype=torch.float32, device=b.device)
    if csc_shape is None:
        return dot_csr_ndarray(out_shape, ones, a_indices, a_indptr, bz)
    return dot_csc_ndarray(csc_shape, tuple(b.shape), ones, a_indices, a_indptr, bz)


def dot_csr_ndarray_sparse(out_shape, a_data, a_indices, a_indptr, b):
    """CSR @ dense returned as CSR (data, indices, indptr[int64]) — `_dot_csr_ndarray_sparse`
    (reference _common.py:758-804) after the GCXS constructor's prune."""
    dense = dot_csr_ndarray(out_shape, a_data, a_indices, a_indptr, b, exact=True)
    mask = _pattern_product(out_shape, a_indices, a_indptr, b.to(dense.dtype) if b.dtype != dense.dtype else b)
    keys, data = _sparsify(dense, mask)
    indptr, indices = keys_to_csr(keys, int(out_shape[0]), int(out_shape[1]), torch.int64)
    return data, indices, indptr


def dot_csc_ndarray_sparse(a_shape, b_shape, a_data, a_indices, a_indptr, b):
    """CSC @ dense returned compressed by COLUMNS — `_dot_csc_ndarray_sparse`
    (reference _common.py:807-866)."""
    M, N = int(a_shape[0]), int(b_shape[1])
    dense = dot_csc_ndarray(a_shape, b_shape, a_data, a_indices, a_indptr, b, exact=True)
    mask = _pattern_product((M, N), a_indices, a_indptr, b.to(dense.dtype) if b.dtype != dense.dtype else b,
                            csc_shape=a_shape)
    keys, data = _sparsify(dense.t().contiguous(), mask.t().contiguous())  # column-major keys: col*M + row
    indptr, indices = keys_to_csr(keys, N, M, torch.int64)
    return data, indices, indptr


def dot_coo_ndarray_sparse(coords, data, b, out_shape, as_keys=False):
    """COO @ dense returned as COO (coords, data) — reference _common.py:1017-1072.  `as_keys`: (sorted linear keys, data)
    instead - the caller's container builds coordinates on demand."""
    dense = dot_coo_ndarray(coords, data, b, out_shape, exact=True)
    keys, vals = _sparsify(dense, numeric=True)
    return (keys, vals) if as_keys else (delinearize(keys, tuple(int(s) for s in out_shape), torch.int64), vals)


def dot_ndarray_coo_sparse(a, coords, data, out_shape, as_keys=False, st=None):
    """dense @ COO returned as COO — reference _common.py:1106-1158."""
    dense = dot_ndarray_coo(a, coords, data, out_shape, exact=True, st=st).contiguous()
    keys, vals = _sparsify(dense, numeric=True)
    return (keys, vals) if as_keys else (delinearize(keys, tuple(int(s) for s in out_shape), torch.int64), vals)


SPGEMM_CHUNK_PRODUCTS = 1 << 27  # products expanded/sorted at a time (bounds workspace to ~5 GB)
SPGEMM_RUN_PROBE_MIN = 1 << 13   # chunks with fewer products: even one output element holding all of them is short work
SPGEMM_RUN_LONG = True           # False: always the one-thread-per-element sums (tests)


def _spgemm_keys(n_row, n_col, a_data, a_indices, a_rows, b_data, b_indices, b_indptr):
    """Expand-sort-compress over row chunks; returns (keys, data) sorted by row*n_col + col with
    every output element summed in the reference's order."""
    from ._reduce import segment_reduce

    dev = require_hip(a_data, b_data)
    dtr = torch_dtype(dot_dtype(a_data.dtype, b_data.dtype))
    vcode = code_of(dtr)
    a_data = a_data.to(dtr).contiguous() if a_data.dtype != dtr else a_data.contiguous()
    b_data = b_data.to(dtr).contiguous() if b_data.dtype != dtr else b_data.contiguous()
    (a_indices, b_indices, b_indptr), it = _unify_index(a_indices.contiguous(), b_indices.contiguous(),
                                                        b_indptr.contiguous())
    nnz_a = int(a_indices.numel())
    s = stream_ptr(dev)
    out_keys, out_vals = [], []
    if nnz_a == 0:
        return torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=dtr, device=dev)
    cnt = torch.empty(nnz_a + 1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spgemm_count", code_of(it), 0, nnz_a, ptr(a_indices), ptr(b_indptr), ptr(cnt), s)
    offs_all = exclusive_scan(cnt)
    total = int(offs_all[-1])
    # chunk boundaries on A elements so that each chunk expands <= SPGEMM_CHUNK_PRODUCTS products;
    # a chunk always ends on a ROW boundary of A so that no output element is split
    p = 0
    row_ends = None
    while p < nnz_a:
        if total - int(offs_all[p]) <= SPGEMM_CHUNK_PRODUCTS:
            q = nnz_a
        else:
            target = int(offs_all[p]) + SPGEMM_CHUNK_PRODUCTS
            q = int(torch.searchsorted(offs_all, torch.tensor([target], device=dev), right=True)[0]) - 1
            q = max(q, p + 1)
            # advance q to the end of the row that element q-1 belongs to
            r = int(a_rows[q - 1])
            q = int(torch.searchsorted(a_rows, torch.tensor([r], device=dev), right=True)[0])
        P = int(offs_all[q]) - int(offs_all[p])
        if P:
            offs = (offs_all[p:q + 1] - offs_all[p]).contiguous()
            keys = torch.empty(P, dtype=torch.int64, device=dev)
            vals = torch.empty(P, dtype=dtr, device=dev)
            _ffi.call("spamd_spgemm_expand", vcode, code_of(it), p, q - p, ptr(a_data), ptr(a_indices), ptr(a_rows),
                      ptr(b_data), ptr(b_indices), ptr(b_indptr), ptr(offs), P, n_col, ptr(keys), ptr(vals), s)
            keys, vals = sort_key_value(keys, vals, max(n_row * n_col - 1, 1))
            heads = flag_heads(keys)
            ho = exclusive_scan(heads)
            c = int(ho[-1])
            summed = None
            if P >= SPGEMM_RUN_PROBE_MIN and P - c >= 1023 and dtr in _CODE_T and dtr != torch.bool:
                # one thread per output element adds its products left to right (the reference's `sums[j] += ...` order, bit for
                # bit) - and walks alone when an element has 5 x 10^4 of them (the diagonal element of a hub row in a @ a.T:
                # 10.5 ms of a 15 ms product, tools/r06/spgemm_hub.py).  When the scan of the head flags shows a window of 1024
                # products without a head (`has_long_run`: P / 1024 words read; P - c products share their element with an
                # earlier one - fewer than a long run's, and there is none), the grouped reduce adds the runs - any length,
                # streaming speed, re-associated: 1e-12 / 1e-5 relative.
                from ._reduce import group_reduce, has_long_run

                if SPGEMM_RUN_LONG and has_long_run(ho):
                    summed = group_reduce(keys, 1, vals, "add", key_bound=max(n_row * n_col, 1), sync=False)[1][:c]
            out_vals.append(summed if summed is not None else segment_reduce(vals, heads, ho, c, "add", sequential=True))
            out_keys.append(compact(keys, heads, ho, c))
        p = q
    if not out_keys:
        return torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=dtr, device=dev)
    return (torch.cat(out_keys), torch.cat(out_vals)) if len(out_keys) > 1 else (out_keys[0], out_vals[0])


SPGEMM_ROW_LOCAL = True  # tuning hook: False = always the global expand-sort-compress


SPGEMM_STATS = {}   # last row-local product: rows, rows left to the global form (diagnostics for the benches)
SPGEMM_BITMAP = True            # wide rows of a matrix with <= 2^20 columns: csrc/spgemm_bitmap.hip (rows written in place)
SPGEMM_BITMAP_MIN_MEAN = 1300   # mean products per row from which the per-row bitmap scan (n_col / 8 bytes of LDS) pays: measured
#                                 at n_col = 10^6, 60000 rows (tools/r04/spgemm_sweep.py; ms buckets / bitmap), round 6 (the bitmap
#                                 kernel of round 5): 400 products per row 1.07 / 2.31, 900: 1.79 / 2.40, 1296: 2.51 / 2.51,
#                                 1600: 2.77 / 2.53, 2025: 3.09 / 2.59, 4096: 6.48 / 3.08, 6400: 8.02 / 3.84, 10^4: 14.0 / 5.34
#                                 (round 4, when 1536 was chosen: 900: 1.91 / 2.55, 2025: 3.36 / 2.89, 10^4: 13.95 / 6.45)
SPGEMM_BITMAP_MAX_DUPS = 120    # expected products per row that share an output element with an earlier one (list of 512)


_BITMAP_UNSUPPORTED = set()    # (device index, split form?) whose launch the library refused once
_BITMAP_REFUSAL_CODES = (1, 2, 9, 701)   # hipError_t values that mean "not with these resources", not "the device faulted"


def _spgemm_bitmap(vcode, it, n_row, n_inner, n_col, parts, total, a_indptr, a_indices, a_data, b_indptr, b_indices, b_data, dtr,
                   dev, s):
    """C = A @ B by csrc/spgemm_bitmap.hip: (data, int64 indices, int64 indptr), or None when a row (or part of a row)
    exceeded the kernel's limits (the caller then tries the next form).  `parts` = 1: whole rows, one workgroup per CU;
    > 1: column ranges, two workgroups per CU.  The result buffers are allocated for every product (an upper bound of the
    result's length) and trimmed: a view when at least 3/4 of them are used."""
    out_idx = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    out_val = torch.empty(max(total, 1), dtype=dtr, device=dev)
    out_ptr = torch.empty(n_row + 1, dtype=torch.int64, device=dev)
    work = torch.empty(n_row * parts + 32, dtype=torch.int64, device=dev)
    bsplit = torch.empty(max(n_inner * (parts - 1), 1), dtype=it, device=dev) if parts > 1 else None
    if (dev.index, parts > 1) in _BITMAP_UNSUPPORTED:
        return None
    try:
        _ffi.call("spamd_spgemm_bitmap", vcode, code_of(it), n_row, n_inner, n_col, parts, ptr(a_indptr), ptr(a_indices), ptr(a_data),
                  ptr(b_indptr), ptr(b_indices), ptr(b_data), ptr(bsplit) if bsplit is not None else None, ptr(work), ptr(out_ptr),
                  ptr(out_idx), ptr(out_val), s)
    except _ffi.HipBackendError as e:
        # the launch itself was REFUSED - a negative SPAMD_E* from the range / LDS checks, or the runtime declining the 160 KB
        # LDS opt-in or the occupancy query on a part with less (hipErrorInvalidValue 1, hipErrorOutOfMemory 2,
        # hipErrorLaunchOutOfResources 701, hipErrorInvalidConfiguration 9): this form is not available on this device -
        # remembered, and the next form (or the bucket kernels) takes the product.  Any other hipError_t is a device fault
        # (an illegal address from a bad operand, a hung queue): never masked, as in _umath._on_device.
        code = getattr(e, "code", 0)
        if code > 0 and code not in _BITMAP_REFUSAL_CODES:
            raise
        _BITMAP_UNSUPPORTED.add((dev.index, parts > 1))
        SPGEMM_STATS["bitmap_refused"] = SPGEMM_STATS.get("bitmap_refused", 0) + 1
        return None
    failed, zeros, nnz = (int(v) for v in torch.cat([work[1:3], out_ptr[-1:]]).tolist())   # ONE read-back
    if os.environ.get("SPAMD_BMK_PROF"):     # (-DBMK_PROF builds of csrc/spgemm_bitmap.hip: cycles per phase, thread 0 of every workgroup)
        SPGEMM_STATS["phase_cycles"] = work[4:20].tolist()
    if failed:
        SPGEMM_STATS["bitmap_failed"] = SPGEMM_STATS.get("bitmap_failed", 0) + 1
        return None
    if nnz * 4 < total * 3:
        out_idx, out_val = out_idx[:nnz].clone(), out_val[:nnz].clone()
    else:
        out_idx, out_val = out_idx[:nnz], out_val[:nnz]
    note_zero_bits_count(out_val, zeros)
    SPGEMM_STATS["rows"], SPGEMM_STATS["heavy_or_declined"], SPGEMM_STATS["kernel"] = n_row, 0, "bitmap"
    SPGEMM_STATS["parts"] = parts
    return out_val, out_idx, out_ptr


SPGEMM_BITMAP_SPLIT = True   # tuning hook: False = only the wide form (whole rows, one workgroup per CU); "first" = the split form first


def _spgemm_bitmap_forms(vcode, n_col, max_prod):
    """The forms of the bitmap kernel to try for this product, best first: a list of `parts` values (1 = wide).  The wide
    form wins where both apply (config-5 share: 13.2 against 14.6 ms with two parts: every part repeats the row's staging,
    scans and look-back with half the threads); the split form is what takes matrices of more than 2^20 columns and rows of
    more than 16384 products."""
    lim = _ffi.lib().spamd_spgemm_bitmap_limits
    forms = []
    if max_prod <= lim(vcode, 0) and n_col <= lim(vcode, 2) and max_prod * max_prod <= 2 * n_col * SPGEMM_BITMAP_MAX_DUPS:
        forms.append(1)
    split_prod, split_cols = int(lim(vcode, 4)), int(lim(vcode, 5))
    if SPGEMM_BITMAP_SPLIT and split_prod and split_cols and (not forms or SPGEMM_BITMAP_SPLIT == "first"):
        parts = max(2, -(-n_col // split_cols))
        # a part holds ~1 / parts of a row's products (uniform columns: + a few standard deviations)
        while parts <= 64 and max_prod / parts + 4 * (max_prod / parts) ** 0.5 + 64 > split_prod:
            parts += 1
        if parts <= 64 and max_prod * max_prod <= 2 * n_col * parts * (SPGEMM_BITMAP_MAX_DUPS // 2):
            forms.insert(0, parts)
    return forms


SPGEMM_SMALL = True
SPGEMM_SMALL_SECOND = True
SPGEMM_SMALL_MAX_CELLS = 1 << 22     # n_row x n_col of the result: its upper-bound buffers (12-16 B per cell) stay below 64 MB
SPGEMM_SMALL_MAX_NNZ = 1 << 16       # stored elements of A (a wave walks its row's elements one after the other): beyond this the
                                     # kernels that spread a row's products over a workgroup have enough work to pay their set-up


SPGEMM_SMALL_MAX_BUFFER = 1 << 30    # entries of the second chance's result buffers (12-16 bytes each)


def _spgemm_small_second(vcode, n_row, n_col, total):
    """Once the row products are known: is this a product for the dense-accumulator kernel after all?  (Round 6,
    tools/r06/spgemm_small_second.py.)  When the result is dense-ish - at least one product per cell of the result - the bucket
    kernel's rows overflow their buckets (a column collects many products) and are redone by the global expand-sort-compress,
    and the bitmap kernel parks too many repeated columns: 3000 x 3000 with 300 elements per row took 28.6 ms, 100000 x 3000
    with 100 per row 91 ms - 1.1 and 7.3 ms with a wave per row over an LDS accumulator, bit-identical.  Below one product
    per cell the accumulator kernel still wins while four waves share a workgroup (rows of at most ~4000 float32 / ~2000
    float64 columns) from a quarter product per cell; with one wave per workgroup or sparser results the other kernels do."""
    if not (SPGEMM_SMALL and SPGEMM_SMALL_SECOND and total and n_row < 2 ** 31):
        return False
    if n_col > int(_ffi.lib().spamd_spgemm_small_max_cols(vcode)) or min(n_row * n_col, total) > SPGEMM_SMALL_MAX_BUFFER:
        return False
    fill = total / (n_row * n_col)
    es = 8 if vcode in (code_of(torch.float64), code_of(torch.int64)) else 4
    four_waves = ((64 * 1024) // 4 - 32) * 8 // (8 * es + 1)       # (sm_max_cols(4), csrc/spgemm_small.hip)
    return fill >= 1.0 or (n_col <= four_waves and fill >= 0.25)




def _spgemm_small(n_row, n_col, a_data, a_indices, a_indptr, b_data, b_indices, b_indptr, bound=None):
    """csrc/spgemm_small.hip: the whole product in one launch and one read-back (the reference's own benchmark sizes,
    benchmarks/test_benchmark_coo.py:9-40, are launch-bound on the general path: 230-300 us against ~70).  None when the
    operands are outside its limits or B is not canonical (the caller takes the general path)."""
    dev = require_hip(a_data, b_data)
    dtr = torch_dtype(dot_dtype(a_data.dtype, b_data.dtype))
    try:
        vcode = code_of(dtr)
    except TypeError:
        return None
    if n_row == 0 or n_col == 0 or n_col > int(_ffi.lib().spamd_spgemm_small_max_cols(vcode)):
        return None     # (a zero-width result: the kernel's argument check refuses n_col <= 0; the general path returns the empty matrix)
    a_data = a_data.to(dtr).contiguous() if a_data.dtype != dtr else a_data.contiguous()
    b_data = b_data.to(dtr).contiguous() if b_data.dtype != dtr else b_data.contiguous()
    (a_indices, a_indptr, b_indices, b_indptr), it = _unify_index(a_indices.contiguous(), a_indptr.contiguous(),
                                                                  b_indices.contiguous(), b_indptr.contiguous())
    cells = n_row * n_col if bound is None else min(n_row * n_col, int(bound))     # (the result holds at most one element per product)
    out_idx = torch.empty(cells, dtype=torch.int64, device=dev)
    out_val = torch.empty(cells, dtype=dtr, device=dev)
    head = torch.empty(2 * n_row + 8, dtype=torch.int64, device=dev)      # [work (n_row + 4) | out_indptr (n_row + 1)]: one buffer, one read-back
    work, out_ptr = head[: n_row + 4], head[n_row + 4: 2 * n_row + 5]
    _ffi.call("spamd_spgemm_small", vcode, code_of(it), n_row, n_col, ptr(a_indptr), ptr(a_indices), ptr(a_data), ptr(b_indptr),
              ptr(b_indices), ptr(b_data), ptr(work), ptr(out_ptr), ptr(out_idx), ptr(out_val), stream_ptr(dev))
    failed, zeros, nnz = read_words(head[1:4])      # ONE read-back (three adjacent words, through pinned host memory)
    if failed:
        return None
    out_idx, out_val = (out_idx[:nnz].clone(), out_val[:nnz].clone()) if nnz * 4 < cells * 3 else (out_idx[:nnz], out_val[:nnz])
    note_zero_bits_count(out_val, zeros)
    SPGEMM_STATS.update(kernel="small", rows=n_row, heavy_or_declined=0)
    return out_val, out_idx, out_ptr


def _spgemm_rows(n_row, n_col, a_data, a_indices, a_indptr, b_data, b_indices, b_indptr):
    """Row-local SpGEMM (csrc/spgemm_rows.hip): (data, int64 indices, int64 indptr) of A @ B.  Rows too heavy for LDS
    are computed by the global expand-sort-compress and merged in; None when most of the work is in such rows (the
    caller then uses the global form throughout)."""
    if SPGEMM_SMALL and n_row * n_col <= SPGEMM_SMALL_MAX_CELLS and int(a_data.numel()) <= SPGEMM_SMALL_MAX_NNZ:
        res = _spgemm_small(n_row, n_col, a_data, a_indices, a_indptr, b_data, b_indices, b_indptr)
        if res is not None:
            return res
    dev = require_hip(a_data, b_data)
    dtr = torch_dtype(dot_dtype(a_data.dtype, b_data.dtype))
    vcode = code_of(dtr)
    if n_col >= 2 ** 31 - 1 or n_row == 0:
        return None
    a_data = a_data.to(dtr).contiguous() if a_data.dtype != dtr else a_data.contiguous()
    b_data = b_data.to(dtr).contiguous() if b_data.dtype != dtr else b_data.contiguous()
    (a_indices, a_indptr, b_indices, b_indptr), it = _unify_index(a_indices.contiguous(), a_indptr.contiguous(),
                                                                  b_indices.contiguous(), b_indptr.contiguous())
    s = stream_ptr(dev)
    prod = torch.empty(n_row + 1, dtype=torch.int64, device=dev)
    maxes = torch.empty(2, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spgemm_row_products", code_of(it), n_row, ptr(a_indptr), ptr(a_indices), ptr(b_indptr), ptr(prod),
              ptr(maxes), s)
    prod_off = exclusive_scan(prod)
    max_prod, max_arow = read_words(maxes) if maxes.dtype == torch.int64 else (int(v) for v in maxes.tolist())
    cap = int(_ffi.lib().spamd_spgemm_rows_capacity(vcode, n_col, max_arow))
    total = int(prod_off[-1])
    lim = _ffi.lib().spamd_spgemm_bitmap_limits
    SPGEMM_STATS.update(max_prod=max_prod, max_arow=max_arow, products=total)
    SPGEMM_STATS.pop("bitmap_failed", None)
    if _spgemm_small_second(vcode, n_row, n_col, total):
        try:
            res = _spgemm_small(n_row, n_col, a_data, a_indices, a_indptr, b_data, b_indices, b_indptr, bound=total)
        except torch.OutOfMemoryError:       # (its buffers hold min(cells, products) entries: the other kernels need less)
            res = None
        if res is not None:
            return res
    if SPGEMM_BITMAP and total and max_arow <= lim(vcode, 1) and total >= SPGEMM_BITMAP_MIN_MEAN * n_row:
        n_inner = int(b_indptr.numel()) - 1
        for parts in _spgemm_bitmap_forms(vcode, n_col, max_prod):
            res = _spgemm_bitmap(vcode, it, n_row, n_inner, n_col, parts, total, a_indptr, a_indices, a_data, b_indptr, b_indices,
                                 b_data, dtr, dev, s)
            if res is not None:
                return res
    SPGEMM_STATS["kernel"] = "buckets"

    def classify(nnz_row):
        """(flags[n_row + 1], rows, their products, rows heavy only by their A length) - csrc/spgemm_rows.hip"""
        flags = new_flags(n_row, dev)
        counts = torch.empty(3, dtype=torch.int64, device=dev)
        _ffi.call("spamd_spgemm_classify_rows", code_of(it), n_row, ptr(prod), ptr(a_indptr), ptr(nnz_row) if nnz_row is not None else 0,
                  cap, ptr(flags), ptr(counts), s)
        return (flags, *[int(v) for v in counts.tolist()])

    if max_prod > cap or max_arow > cap:
        # rows too heavy for LDS (products, or A elements to stage) go through the global expand-sort-compress
        _, n_heavy, heavy_products, n_arow_only = classify(None)
        if n_heavy * 4 > n_row or heavy_products * 4 > total * 3 or n_arow_only:
            return None  # mostly heavy (or rows heavy only by their A length): the global form throughout
    tmp_cols = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    tmp_vals = torch.empty(max(total, 1), dtype=dtr, device=dev)
    nnz_row = torch.zeros(n_row + 1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spgemm_rows", vcode, code_of(it), n_row, n_col, ptr(a_indptr), ptr(a_indices), ptr(a_data),
              ptr(b_indptr), ptr(b_indices), ptr(b_data), ptr(prod_off), min(max_prod, cap), max_arow, ptr(tmp_cols),
              ptr(tmp_vals), ptr(nnz_row), s)
    # rows above the capacity were skipped, rows with a column of more than 64 products were declined (nnz_row = -1):
    # both are computed by the global form and copied into the scratch
    flags, n_heavy, _, _ = classify(nnz_row)
    SPGEMM_STATS["rows"], SPGEMM_STATS["heavy_or_declined"] = n_row, n_heavy
    if n_heavy:
        iota = torch.empty(max(n_row, int(a_indices.numel())) + 1, dtype=torch.int64, device=dev)
        _ffi.call("spamd_iota", int(iota.numel()), ptr(iota), s)
        heavy_rows = compact(iota[:n_row], flags, exclusive_scan(flags), n_heavy)
        a_rows = csr_to_keys(a_indptr, torch.zeros_like(a_indices), n_row, 1)   # row id of every A element
        nA = int(a_indices.numel())
        eflags = new_flags(nA, dev)
        _ffi.call("spamd_gather", 8, nA, ptr(flags), ptr(a_rows), ptr(eflags), s)
        eflags[nA:] = 0
        eoff = exclusive_scan(eflags)
        sel = compact(iota[:nA], eflags, eoff, int(eoff[-1]))               # A elements of those rows
        hkeys, hvals = _spgemm_keys(n_row, n_col, gather(a_data, sel), gather(a_indices, sel), gather(a_rows, sel), b_data,
                                    b_indices, b_indptr)
        hptr, hidx = keys_to_csr(hkeys, n_row, n_col, torch.int64)
        _ffi.call("spamd_spgemm_unpack", vcode, n_heavy, ptr(heavy_rows.contiguous()), ptr(hptr),
                  ptr(hidx), ptr(hvals.contiguous()), ptr(prod_off), ptr(tmp_cols), ptr(tmp_vals), ptr(nnz_row), s)
    out_ptr = exclusive_scan(nnz_row)
    nnz = int(out_ptr[-1])
    out_idx = torch.empty(nnz, dtype=torch.int64, device=dev)
    out_val = torch.empty(nnz, dtype=dtr, device=dev)
    zeros = torch.empty(1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spgemm_pack", vcode, n_row, ptr(prod_off), ptr(out_ptr), ptr(tmp_cols), ptr(tmp_vals), ptr(out_idx),
              ptr(out_val), ptr(zeros), s)
    note_zero_bits_count(out_val, zeros)
    return out_val, out_idx, out_ptr


def dot_csr_csr(out_shape, a_data, b_data, a_indices, b_indices, a_indptr, b_indptr):
    """CSR @ CSR -> (data, indices, indptr) with int64 indices — `_dot_csr_csr`
    (reference _common.py:639-717).  Explicit zeros are kept (the GCXS constructor prunes)."""
    n_row, n_col = int(out_shape[0]), int(out_shape[1])
    if SPGEMM_ROW_LOCAL:
        res = _spgemm_rows(n_row, n_col, a_data, a_indices, a_indptr, b_data, b_indices, b_indptr)
        if res is not None:
            return res
    a_rows = csr_to_keys(a_indptr, torch.zeros_like(a_indices), n_row, 1)  # row id of every A element
    keys, data = _spgemm_keys(n_row, n_col, a_data, a_indices, a_rows, b_data, b_indices, b_indptr)
    indptr, indices = keys_to_csr(keys, n_row, n_col, torch.int64)
    return data, indices, indptr


def dot_coo_coo(out_shape, a_coords, b_coords, a_data, b_data, n_inner):
    """COO @ COO -> (coords[2, nnz] int64, data), sorted — `_dot_coo_coo` + the indptr
    construction of reference _common.py:450-475,907-976.  `n_inner` = a.shape[1] = b.shape[0]."""
    n_row, n_col = int(out_shape[0]), int(out_shape[1])
    b_indptr = rows_to_indptr(b_coords[0], int(n_inner))
    if SPGEMM_ROW_LOCAL:
        a_indptr = rows_to_indptr(a_coords[0], n_row)
        res = _spgemm_rows(n_row, n_col, a_data, a_coords[1], a_indptr, b_data, b_coords[1], b_indptr)
        if res is not None:
            data, indices, indptr = res
            keys = csr_to_keys(indptr, indices, n_row, n_col)
            return delinearize(keys, (n_row, n_col), torch.int64), data
    a_rows = convert(a_coords[0].contiguous(), torch.int64)
    keys, data = _spgemm_keys(n_row, n_col, a_data, a_coords[1], a_rows, b_data, b_coords[1], b_indptr)
    return delinearize(keys, (n_row, n_col), torch.int64), data


SDDMM_PANEL_BYTES = 3 << 20     # Bt rows of one column panel: what stays in a 4 MiB L2 next to the streamed operands
SDDMM_PANEL_MIN_NNZ = 1 << 18   # below this the kernel is launch-bound either way
SDDMM_PANEL_GAIN = 0.75         # share of a gathered Bt row that the panel order saves (the rest: scattered output, streams)


class SddmmPanels:
    """Column-panel order of a mask, or of a subset of its stored elements (csrc/sddmm.hip, spamd_sddmm_panels):
    `pos` = the elements' positions in the mask, stably sorted by column panel; `rows`/`cols` = their coordinates in
    that order.  Depends on the coordinates and the panel width only: cached on the mask by `sparse_amd.sddmm`."""

    __slots__ = ("pos", "rows", "cols", "width", "count", "nnz", "chunk", "xstate", "xmax", "_vals", "_vals_key", "row_runs")

    def values(self, s_orig, s_data):
        """`s_data` (= `s_orig` in the accumulation dtype) in panel order; kept for as long as the same, unmodified
        `s_orig` is passed."""
        key = (s_orig.data_ptr(), s_orig._version, s_orig.dtype, s_data.dtype)
        if self._vals_key != key:
            self._vals, self._vals_key = gather(s_data, self.pos), key
        return self._vals


def sddmm_has_panels(dtype, Kd):
    """Does spamd_sddmm_panels have a kernel for rows of `Kd` elements of `dtype`?"""
    return dtype in (torch.bfloat16, torch.float32, torch.float64) and bool(_ffi.lib().spamd_sddmm_has_panels(code_of(dtype), int(Kd)))


SDDMM_PAD_MIN_NNZ = 200_000     # samples from which padding the inner dimension to a row-cached kernel's row length pays the two copies
_SDDMM_ROW_BYTES = (256, 512, 768, 1024, 1536, 2048, 3072, 4096)      # rows the row-cached / panel kernels are instantiated for


def sddmm_pad_inner(a, bt, nnz):
    """(a, bt) with the inner dimension zero-padded to the next row length the row-cached kernels have, when that is at most
    1.5x the rows' own length (round 6, tools/r06/sddmm_k_sweep.py: K = 96 or 100 in float32, 192 in bfloat16 - 384-byte rows -
    ran through the generic gather, 0.61 / 0.85 ms at config 4's mask, without a panel order; as 512-byte rows 0.38 / 0.33 ms
    plus two copies of ~0.03 ms).  Zeros add nothing to a dot product: the sums are those of the padded kernels' lane order,
    in the mask's own order and in panel order alike."""
    if a.dtype not in (torch.bfloat16, torch.float32, torch.float64) or a.dim() != 2 or nnz < SDDMM_PAD_MIN_NNZ:
        return a, bt
    esz, k = a.element_size(), int(a.shape[1])
    if k == 0 or sddmm_has_panels(a.dtype, k):
        return a, bt
    rb = -(-k * esz // 16) * 16
    want = next((w for w in _SDDMM_ROW_BYTES if w >= rb), None)
    if want is None or want * 2 > rb * 3:
        return a, bt
    pad = want // esz - k
    return torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(bt, (0, pad))


def sddmm_panel_width(bt):
    """Bt rows per panel, or 0 when Bt fits the L2 as a whole or its K has no row-cached kernel (no panel order)."""
    row_bytes = int(bt.shape[1]) * bt.element_size()
    if row_bytes == 0 or int(bt.shape[0]) * row_bytes <= SDDMM_PANEL_BYTES:
        return 0
    if not sddmm_has_panels(bt.dtype, bt.shape[1]):
        return 0
    if SDDMM_TWO_PASS:
        # rows of 1 KB run as two passes over 512-byte halves (round 5): a panel is sized for the half-rows
        row_bytes = int(_ffi.lib().spamd_sddmm_panel_row_bytes(code_of(bt.dtype), int(bt.shape[1]))) or row_bytes
    if SDDMM_XCD_PANELS and int(bt.shape[0]) * row_bytes >= 8 * SDDMM_PANEL_BYTES:
        # XCD-private panels: a multiple of eight panels of at most 7/6 of the panel size (3.5 MiB), so that every XCD owns the same number
        # (measured at config 4: 16 panels of 6250 rows 0.357 ms private vs 0.390 shared; 17 panels of 6144 rows 0.402 vs 0.396)
        per_xcd = -(-int(bt.shape[0]) * row_bytes // (8 * (SDDMM_PANEL_BYTES * 7 // 6)))
        return max(-(-int(bt.shape[0]) // (8 * per_xcd)), 64)
    return max(SDDMM_PANEL_BYTES // row_bytes, 64)


def sddmm_panels_pay(n, a, bt, width):
    """Column-panel order or the mask's own order for `n` stored elements?  In the mask's order every element fetches
    its Bt row (K * itemsize bytes) through the fabric; in panel order the Bt rows come from the L2, but A is streamed
    once per panel and Bt once per XCD, whatever n is (measured at config 4's shapes: 10^7 elements 0.40 vs 0.70 ms,
    3 * 10^6 elements 0.33 vs 0.21 ms)."""
    if not width or n < SDDMM_PANEL_MIN_NNZ:
        return False
    row_bytes = int(bt.shape[1]) * bt.element_size()
    panels = -(-int(bt.shape[0]) // int(width))
    fixed = panels * int(a.shape[0]) * row_bytes + 8 * int(bt.shape[0]) * row_bytes
    return n * row_bytes * SDDMM_PANEL_GAIN > fixed


SDDMM_TILE_EQUIV = 410      # a 32 x 32 tile product costs what the sampled kernel spends on this many samples INSIDE a dense tile
SDDMM_REST_PENALTY = 2.5    # left-over samples per sample gained that the split may cost (they leave the panel order)


def sddmm_tiles_pay(plan, a, bt, width):
    """Dense tiles on the matrix cores + the rest sampled, or everything sampled?  Measured at K = 256 bf16
    (tools/sddmm_crossover.py, bench_paths A9_mfma_*): a tile product takes ~9 ns whatever the tile holds, the sampled
    kernel ~0.022 ns per sample inside a dense tile (A and Bt rows are shared), so the tiles gain
    n_dense - 410 * ntiles samples' worth of time; the left-over samples, run on their own, cost ~0.10 instead of
    ~0.045 ns each when the whole mask would have taken the panel order (both scale with the row length alike)."""
    ntiles = int(plan.tiles.numel())
    gain = plan.n_dense_samples - SDDMM_TILE_EQUIV * ntiles
    if ntiles == 0 or gain <= 0:
        return False
    nrest = int(plan.rest.numel())
    if sddmm_panels_pay(plan.nnz, a, bt, width) and not sddmm_panels_pay(nrest, a, bt, width):
        return gain > SDDMM_REST_PENALTY * nrest
    return True


SDDMM_TWO_PASS = True     # rows of exactly 1 KB in two launches over 512-byte halves and half-row panels (False: the row-cached kernel on whole rows, rounds 2-4)
SDDMM_XCD_PANELS = True   # panels are private to an XCD (workgroup b serves XCD b % 8: the placement MI355X is observed to use;
                          # only speed depends on it); False: every XCD walks every panel


def sddmm_panels(coords, shape, width, subset=None, xcd=None):
    """Panel order of the mask's stored elements, or of those listed in `subset` (int64 positions, ascending).  With `xcd`
    (default SDDMM_XCD_PANELS, and only when there are at least 8 panels) the order is XCD-major: panel p belongs to XCD
    p % 8, and `xstate` holds where each XCD's elements start."""
    dev = require_hip(coords)
    rows, cols = coords[0].contiguous(), coords[1].contiguous()
    if not index_dtype_ok(rows):
        rows, cols = rows.to(torch.int64), cols.to(torch.int64)
    p = SddmmPanels()
    p.nnz = int(rows.numel())
    if subset is not None:
        rows, cols = gather(rows, subset), gather(cols, subset)
    n = int(rows.numel())
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    npanels = (int(shape[1]) - 1) // int(width) + 1
    xcd = SDDMM_XCD_PANELS if xcd is None else xcd
    per_xcd = -(-npanels // 8) if xcd and npanels >= 8 else 0
    _ffi.call("spamd_sddmm_panel_keys", code_of(cols.dtype), n, ptr(cols), int(width), per_xcd, ptr(keys), stream_ptr(dev))
    skeys, perm = sort_keys(keys, max(8 * per_xcd - 1 if per_xcd else npanels - 1, 1))
    p.pos = perm if subset is None else gather(subset, perm)
    p.rows, p.cols, p.width, p.count, p.chunk = gather(rows, perm), gather(cols, perm), int(width), n, 0
    p.xstate, p.xmax = None, 0
    # runs of equal rows in the panel order (one small read-back at plan time): 256 consecutive elements hold about
    # 256 * row_runs / count + 1 distinct A rows - what the panel kernel's LDS slots are sized for (`_sddmm_row_slots`)
    p.row_runs = int((p.rows[1:] != p.rows[:-1]).sum().item()) + 1 if n > 0 else 0
    if per_xcd:
        # first element of every (XCD, panel) key in the sorted order -> the nine boundaries of the XCDs' ranges
        first = rows_to_indptr(skeys, 8 * per_xcd)[::per_xcd].contiguous()
        host = first.cpu().numpy()                       # (plan time: one small read-back for the launch grid)
        p.xstate, p.xmax = first, int(np.max(np.diff(host)))
    p._vals = p._vals_key = None
    return p


def _sddmm_row_slots(panels, row_bytes, idx_bytes):
    """LDS slots for A rows per 256-element workgroup of the panel kernel (`chunk` of spamd_sddmm_panels): enough for the
    distinct rows such a workgroup meets - a row without a slot is fetched from memory and waited for on the spot - but not
    past three workgroups per CU.  (Round 6, tools/r06/sddmm_cap_sweep.py at config 4's mask: 768-byte rows, 61 distinct
    rows per workgroup: 32 slots - the kernel's own default, 24 KB - 0.82 ms, 52: 1.00, 64: 0.59, 72-96: 0.64, 128: 1.07;
    512-byte rows, 41 distinct: 44-72 slots 0.324, 80-96: 0.356; 256-byte rows, 21 distinct: 48-72 slots 0.172, 96 - the
    default - 0.199.)"""
    if panels.chunk or not panels.count or row_bytes >= 1024:
        return int(panels.chunk)
    expected = 256.0 * panels.row_runs / panels.count + 1.0
    default = max(16, (24 << 10) // row_bytes)                # (the kernel's own choice: 24 KB of rows, launch_panel)
    three = (160 * 1024 // 3 - 2048 - 1024 - 2 * 256 * idx_bytes - 16) // row_bytes    # (- 2 KB: LDS is handed out in blocks; 65 slots of 768 bytes already leave two workgroups per CU)
    if expected + 2 > default:                                 # rows without a slot would be the rule
        return max(16, min(int(1.3 * expected) + 8, three, 256))
    if default > 64 and 1.3 * expected + 8 <= 64:              # short rows: 64 slots instead of 96 (0.172 against 0.199 ms)
        return 64
    return 0


def _sddmm_panels_into(panels, s_orig, s_data, a, bt, out):
    """The elements of `panels` (all of the mask or a subset), written to their positions in `out`."""
    part = None
    row_bytes = int(a.shape[1]) * a.element_size()
    if SDDMM_TWO_PASS and row_bytes == 1024:
        part = torch.empty(panels.count, dtype=out.dtype, device=out.device)     # (first-half sums, panel order)
        row_bytes = 512
    slots = _sddmm_row_slots(panels, row_bytes, panels.rows.element_size())
    _ffi.call("spamd_sddmm_panels", code_of(a.dtype), code_of(out.dtype), code_of(panels.rows.dtype), panels.count,
              ptr(panels.rows), ptr(panels.cols), ptr(panels.pos), ptr(panels.values(s_orig, s_data)), ptr(a), a.stride(0),
              ptr(bt), bt.stride(0), int(a.shape[1]), slots, ptr(panels.xstate) if panels.xstate is not None else None,
              int(panels.xmax), ptr(part) if part is not None else None, ptr(out), stream_ptr(out.device))


def sddmm_coo(coords, s_data, a, bt, panels=None):
    """out[n] = s[n] * <a[i_n, :], bt[j_n, :]> for a 2-D COO mask — the sampled product the
    reference writes as `s * (a @ b)` (examples/sddmm_example.py:51-52).  `panels` (SddmmPanels of the same
    mask) selects the column-panel order."""
    dev = require_hip(coords, s_data, a, bt)
    nnz = int(s_data.numel())
    if a.dtype != bt.dtype:
        raise TypeError("a and bt must share a dtype")
    if a.dtype == torch.bfloat16 or a.dtype == torch.float32:
        sdt = torch.float32
    elif a.dtype == torch.float64:
        sdt = torch.float64
    else:
        raise TypeError(f"sddmm supports bfloat16/float32/float64 dense operands, got {a.dtype}")
    s_orig, s_data = s_data, s_data.to(sdt).contiguous()
    a, bt = a.contiguous(), bt.contiguous()
    if a.shape[1] != bt.shape[1]:
        raise ValueError("shape-mismatch for sum")
    esz = a.element_size()
    if (a.shape[1] * esz) % 16:
        pad = (16 - (a.shape[1] * esz) % 16) // esz  # row pitch must be 16-byte aligned
        a = torch.nn.functional.pad(a, (0, pad))
        bt = torch.nn.functional.pad(bt, (0, pad))
    if a.data_ptr() % 16:      # the kernels load 16-byte vectors: a view at an odd storage offset is copied once
        a = a.clone()
    if bt.data_ptr() % 16:
        bt = bt.clone()
    rows, cols = coords[0].contiguous(), coords[1].contiguous()
    if not index_dtype_ok(rows):
        rows, cols = rows.to(torch.int64), cols.to(torch.int64)
    out = torch.empty(nnz, dtype=sdt, device=dev)
    if panels is not None and nnz and panels.nnz == nnz and panels.count == nnz and sddmm_has_panels(a.dtype, a.shape[1]):
        _sddmm_panels_into(panels, s_orig, s_data, a, bt, out)
        return out
    _ffi.call("spamd_sddmm", code_of(a.dtype), code_of(sdt), code_of(rows.dtype), nnz, ptr(rows), ptr(cols),
              ptr(s_data), ptr(a), a.stride(0), ptr(bt), bt.stride(0), int(a.shape[1]), ptr(out), stream_ptr(dev))
    return out


SDDMM_TILE_THRESHOLD = 512     # samples per 32 x 32 mask tile from which the matrix-core tile product is clearly the cheaper one (tools/sddmm_crossover.py, profiles/r02_sddmm_crossover.txt: 1.5x at 512, 2.3x at 1024, break-even ~150 against the panel-order sampled kernel but ~400 against the sampled kernel on a dense tile)
SDDMM_MFMA_MIN_SHARE = 0.05    # (sddmm_coo_mfma without `force`) below this share of samples in dense tiles it declines


class SddmmPlan:
    """Per-mask dispatch between the matrix-core tile kernel and the sampled kernel (csrc/sddmm_mfma.hip):
    `tiles` = runs (in `seg_start`) of the tiles that take the MFMA path, `rest` = sample indices left to the sampled
    kernel.  Depends on the coordinates only: cached on the mask by `sparse_amd.sddmm`."""

    __slots__ = ("keys", "perm", "seg_start", "tiles", "rest", "tile_cols", "n_dense_samples", "nnz", "threshold")


def sddmm_plan(coords, shape, threshold=None):
    dev = require_hip(coords)
    threshold = SDDMM_TILE_THRESHOLD if threshold is None else int(threshold)
    rows, cols = coords[0].contiguous(), coords[1].contiguous()
    if not index_dtype_ok(rows):
        rows, cols = rows.to(torch.int64), cols.to(torch.int64)
    nnz = int(rows.numel())
    ts = int(_ffi.lib().spamd_sddmm_tile_size())
    tile_rows, tile_cols = -(-int(shape[0]) // ts), -(-int(shape[1]) // ts)
    p = SddmmPlan()
    p.nnz, p.tile_cols, p.threshold = nnz, tile_cols, threshold
    s = stream_ptr(dev)
    keys = torch.empty(nnz, dtype=torch.int64, device=dev)
    _ffi.call("spamd_sddmm_tile_keys", code_of(rows.dtype), nnz, ptr(rows), ptr(cols), tile_cols, ptr(keys), s)
    p.keys, p.perm = sort_keys(keys, max(tile_rows * tile_cols - 1, 1))
    heads = flag_heads(p.keys)
    hoff = exclusive_scan(heads)
    nseg = int(hoff[-1])
    iota = torch.empty(nnz + 1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_iota", nnz + 1, ptr(iota), s)
    seg_start = torch.empty(nseg + 1, dtype=torch.int64, device=dev)
    if nseg:
        _ffi.call("spamd_compact", 8, nnz, ptr(iota), ptr(heads), ptr(hoff), ptr(seg_start), s)
    seg_start[nseg:] = nnz   # (one 8-byte fill: the end of the last run)
    p.seg_start = seg_start
    tflag, sflag = new_flags(nseg, dev), new_flags(nnz, dev)
    _ffi.call("spamd_sddmm_tile_classify", nseg, ptr(seg_start), threshold, ptr(tflag), ptr(sflag), s)
    toff, soff = exclusive_scan(tflag), exclusive_scan(sflag)
    ntile, nrest = int(toff[-1]), int(soff[-1])
    p.tiles = compact(iota[:nseg].contiguous(), tflag, toff, ntile) if nseg else torch.empty(0, dtype=torch.int64, device=dev)
    p.rest = compact(p.perm, sflag, soff, nrest)
    p.n_dense_samples = nnz - nrest
    return p


def sddmm_coo_mfma(plan, coords, shape, s_data, a, bt, out=None, force=False, rest_panels=None):
    """SDDMM with per-tile dispatch (bf16 operands): dense tiles on the matrix cores, the rest through the sampled
    kernel (in column-panel order when `rest_panels` = sddmm_panels(..., subset=plan.rest) is given).  Returns None
    when the plan leaves (almost) everything to the sampled kernel and `force` is not set."""
    dev = require_hip(coords, s_data, a, bt)
    if a.dtype != torch.bfloat16 or bt.dtype != torch.bfloat16:
        raise TypeError("the matrix-core SDDMM path takes bfloat16 operands")
    Kd = int(a.shape[1])
    if Kd % 16 or Kd == 0:
        return None
    if not force and plan.n_dense_samples < SDDMM_MFMA_MIN_SHARE * plan.nnz:
        return None
    a, bt = a.contiguous(), bt.contiguous()
    if a.data_ptr() % 16 or bt.data_ptr() % 16:
        return None     # the tile kernel loads 16-byte operand fragments (a view at an odd storage offset): sampled kernel
    s_orig, s_data = s_data, s_data.to(torch.float32).contiguous()
    rows, cols = coords[0].contiguous(), coords[1].contiguous()
    if not index_dtype_ok(rows):
        rows, cols = rows.to(torch.int64), cols.to(torch.int64)
    if out is None:
        out = torch.empty(plan.nnz, dtype=torch.float32, device=dev)
    s = stream_ptr(dev)
    _ffi.call("spamd_sddmm_mfma_tiles", code_of(rows.dtype), int(plan.tiles.numel()), ptr(plan.tiles), ptr(plan.seg_start),
              ptr(plan.keys), ptr(plan.perm), plan.tile_cols, int(shape[0]), int(shape[1]), ptr(rows), ptr(cols), ptr(s_data),
              ptr(a), a.stride(0), ptr(bt), bt.stride(0), Kd, ptr(out), s)
    nrest = int(plan.rest.numel())
    if nrest and rest_panels is not None and rest_panels.count == nrest and rest_panels.nnz == plan.nnz and \
            sddmm_has_panels(a.dtype, Kd):
        _sddmm_panels_into(rest_panels, s_orig, s_data, a, bt, out)
    elif nrest:
        sub = torch.empty(nrest, dtype=torch.float32, device=dev)
        rr, cc, ss = gather(rows, plan.rest), gather(cols, plan.rest), gather(s_data, plan.rest)
        _ffi.call("spamd_sddmm", code_of(a.dtype), code_of(torch.float32), code_of(rr.dtype), nrest, ptr(rr), ptr(cc), ptr(ss),
                  ptr(a), a.stride(0), ptr(bt), bt.stride(0), Kd, ptr(sub), s)
        scatter_into(out, plan.rest, sub)
    return out


# ---------------------------------------------------------------------------------------------
# inspector/executor SpMM (csrc/spmm_tiled.hip)
# ---------------------------------------------------------------------------------------------
def tiled_params(dtype=torch.float32):
    """(rows per group, B rows per tile, groups per workgroup, entries per block, slack blocks, max tiles of the
    direct inspector, columns per panel) for float32 / float64 values."""
    v = [_ct.c_int(0) for _ in range(7)]
    _ffi.call("spamd_spmm_tiled_params", code_of(_tiled_layout_dtype(torch_dtype(dtype))), *[_ct.byref(x) for x in v])
    return tuple(x.value for x in v)


TILED_DTYPES = (torch.float32, torch.float64, torch.int32)


def _tiled_layout_dtype(dtype):
    """int32 values travel in the float32 layout (the inspector only moves value bits; the executor's int32 variant does
    the arithmetic: SPAMD_TILED_INT32)"""
    return torch.float32 if dtype == torch.int32 else dtype

_TOUCH_OVERRIDE = int(__import__("os").environ.get("SPARSE_AMD_TILED_TOUCH", "0"))   # tuning hook: lines prefetched per list


TILED_ONE_PASS_INSPECTOR = True   # False: the two-pass builder (count, scan, fill) - kept for cross-checks


class TiledLayout(tuple):
    """(blocks, blk_off, value dtype) + `mean_blocks` = mean 64-byte blocks per (row group, tile) list, which the
    launcher turns into the executor's prefetch width without reading anything back from the device, + `group_ends`:
    blk_off holds tiles + 1 entries per row group (the one-pass inspector's form) instead of one running array"""

    def __new__(cls, blocks, blk_off, dtype, mean_blocks, group_ends=False, pending=None, rowmap=None, groups=None):
        self = super().__new__(cls, (blocks, blk_off, dtype))
        self.mean_blocks = mean_blocks
        self.group_ends = group_ends
        self.pending = pending   # device word of a one-pass layout whose "unsorted column indices" verdict was not read yet
        self.rowmap = rowmap     # balanced layout (round 5): the row of every slot of every group, int32[groups * rows_per_group + 16]
        self.groups = groups
        return self


class UnsortedColumns(ValueError):
    """A one-pass tiled layout built with `defer_check=True` turned out to come from rows with unsorted column indices:
    the layout (and the product just computed from it) is invalid; rebuild with `force_sort=True`."""


def csr_tiled_layout(a_data, a_indices, a_indptr, M, Kd, force_sort=False, dtype=None, defer_check=False, balance=None):
    """Inspector: the K-tiled block stream of a CSR matrix used by `dot_csr_ndarray_tiled`, for float32 or float64
    values (`dtype`, default: a_data's if it is one of them, else float32).
    Returns (blocks int32[(total_blocks + slack) * 16], blk_off int32[nseg + 1], value dtype).
    Sorted column indices and a moderate K take the direct one-pass builder (count + fill in one launch, row groups
    independent of each other; the stream is then allocated for its upper bound, groups start at closed-form offsets
    with zeroed blocks between them, and blk_off is int32[groups * (tiles + 1)]: `TiledLayout.group_ends`);
    anything else the general key-sort recipe.  The one-pass builder reports unsorted column indices in a device word;
    with `defer_check` that word is not read here (no host wait between the inspector and the first product):
    `dot_csr_ndarray_tiled` reads it after enqueueing its first product and raises `UnsortedColumns`.
    `balance` (round 5, skewed row lengths): None = decide - with `defer_check` the test itself is deferred as well (the natural
    layout is built, the heaviest row group's size travels with the "unsorted" word and is read behind the first product,
    which then marks the layout `rebalance` for its owner: no host wait on the common, unskewed path), without it the
    statistics are read here; True = build the balanced layout if the operand is skewed at all, reading here; False = never."""
    dev = require_hip(a_data, a_indices, a_indptr)
    if dtype is None:
        dtype = a_data.dtype if a_data.dtype in TILED_DTYPES else torch.float32
    dtype = torch_dtype(dtype)
    vc = code_of(_tiled_layout_dtype(dtype))
    rg, kb, gpb, epb, slack, direct_max, _ = tiled_params(dtype)
    nnz = int(a_data.numel())
    ntiles = -(-Kd // kb)
    groups = -(-(-(-M // rg)) // gpb) * gpb
    nseg = groups * ntiles
    s = stream_ptr(dev)
    vals = a_data.to(dtype).contiguous()
    if dtype == torch.int32:
        vals = vals.view(torch.float32)
    if not index_dtype_ok(a_indices) or a_indices.dtype != a_indptr.dtype:
        a_indices, a_indptr = a_indices.to(torch.int64), a_indptr.to(torch.int64)

    def finish(blk_off, fill):
        total = int(blk_off[-1])
        if total >= 2 ** 31:
            raise ValueError("tiled SpMM layout: more than 2^31 blocks")
        blocks = torch.empty((total + slack) * 16, dtype=torch.int32, device=dev)
        fill(blk_off, total, blocks)
        return TiledLayout(blocks, convert(blk_off, torch.int32), dtype, total / max(nseg, 1))

    may_balance = (ntiles <= direct_max and not force_sort and TILED_ONE_PASS_INSPECTOR and TILED_BALANCE and balance is not False
                   and nnz >= TILED_BALANCE_MIN_NNZ and 0 < M < 2 ** 31)
    if may_balance and (balance or not defer_check):
        lay = _balanced_tiled_layout(vals, a_indices.contiguous(), a_indptr.contiguous(), M, Kd, dtype, nnz, defer_check, dev, s)
        if lay is not None:
            return lay
        may_balance = False      # (measured here: not skewed)
    if ntiles <= direct_max and not force_sort and TILED_ONE_PASS_INSPECTOR:
        # one pass over A (csrc/spmm_tiled.hip `tl_inspect_kernel`): count, fill and padding in a single launch with no
        # dependence between row groups (a group's first block is a closed-form upper bound from the row pointers); the
        # stream is allocated for its upper bound, so nothing waits for a total
        ic = code_of(a_indices.dtype)
        ind, ptr_ = a_indices.contiguous(), a_indptr.contiguous()
        upper = -(-nnz // epb) + nseg
        if upper < 2 ** 31:
            blocks = torch.empty((upper + slack) * 16, dtype=torch.int32, device=dev)
            blk_off = torch.empty(groups * (ntiles + 1), dtype=torch.int32, device=dev)
            state = torch.empty(9, dtype=torch.int64, device=dev)       # [0] unsorted columns; [1..8] = the skew statistics' eight words ([7] = heaviest natural group)
            _ffi.call("spamd_spmm_tiled_inspect", vc, ic, M, Kd, ptr(vals), ptr(ind), ptr(ptr_), ptr(state), ptr(blk_off),
                      ptr(blocks), s)
            if defer_check:
                lay = TiledLayout(blocks, blk_off, dtype, nnz / epb / max(nseg, 1) + 0.5, group_ends=True, pending=state)
                if may_balance:
                    # the skew test rides along: the heaviest natural group (one thread per group) lands in state[7]
                    _ffi.call("spamd_spmm_tiled_map_stats", ic, M, Kd, 32, ptr(ptr_), None, None, ptr(state) + 8, s)
                    lay.skew_mean = nnz * rg / max(M, 1)
                return lay
            if int(state[0]) == 0:      # (sorted column indices everywhere)
                return TiledLayout(blocks, blk_off, dtype, nnz / epb / max(nseg, 1) + 0.5, group_ends=True)
    elif ntiles <= direct_max and not force_sort:
        nblk = torch.empty(nseg + 1, dtype=torch.int64, device=dev)
        flags = torch.empty(1, dtype=torch.int32, device=dev)
        ic = code_of(a_indices.dtype)
        ind, ptr_ = a_indices.contiguous(), a_indptr.contiguous()
        _ffi.call("spamd_spmm_tiled_count", vc, ic, M, Kd, ptr(ind), ptr(ptr_), ptr(nblk), ptr(flags), s)
        blk_off = exclusive_scan(nblk)
        if int(flags[0]) == 0:
            return finish(blk_off, lambda bo, total, blocks: _ffi.call(
                "spamd_spmm_tiled_fill", vc, ic, M, Kd, ptr(vals), ptr(ind), ptr(ptr_), ptr(bo), total, ptr(blocks), s))
    rc = csr_to_keys(a_indptr, a_indices, M, Kd)
    tk = torch.empty_like(rc)
    _ffi.call("spamd_spmm_tiled_keys", nnz, ptr(rc), Kd, ptr(tk), s)
    del rc
    tk, vals = sort_key_value(tk, vals, max(nseg * rg * kb - 1, 1))
    seg_start = torch.empty(nseg + 1, dtype=torch.int64, device=dev)
    nblk = torch.empty(nseg + 1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spmm_tiled_lists", vc, nnz, ptr(tk), M, Kd, ptr(seg_start), ptr(nblk), s)
    return finish(exclusive_scan(nblk), lambda bo, total, blocks: _ffi.call(
        "spamd_spmm_tiled_pack", vc, nnz, ptr(tk), ptr(vals), ptr(seg_start), ptr(bo), total, ptr(blocks), s))


TILED_BALANCE = True            # skewed row lengths: a balanced (row-mapped) layout instead of consecutive rows per wave
TILED_BALANCE_MIN_NNZ = 1 << 20
TILED_BALANCE_SKEW = 1.6        # natural layout kept while its heaviest row group holds at most this x the mean group
TILED_BALANCE_CAPMUL = 2        # cap = the power of two >= this x the mean 35-row group (rows above cap / 2 get a group of their own)
TILED_BALANCE_STATS = {}        # the last decision (tests, benches)


def _balanced_tiled_layout(vals, ind, ptr_, M, Kd, dtype, nnz, defer_check, dev, s):
    """The block stream of a SKEWED matrix (csrc/spmm_tiled.hip, `tl_map_*`): rows sorted by length, heavy rows given
    groups of their own (or shared by 2 / 4 / 8 / 16), a workgroup's 16 groups equally heavy, rows read and stored
    through a row map.  Returns None when the natural layout is balanced enough (its heaviest 35-row group at most
    TILED_BALANCE_SKEW x the mean: uniform `random` matrices measure ~1.1) - one small read-back decides.  Measured on a Zipf
    matrix of config 2's size (bench row A1_powerlaw): natural layout 3.5 ms per product, balanced see DESIGN.md."""
    rg, kb, gpb, epb, slack, direct_max, _ = tiled_params(dtype)
    ntiles = -(-Kd // kb)
    ic, vc = code_of(ind.dtype), code_of(_tiled_layout_dtype(dtype))
    mean_group = nnz * rg / max(M, 1)
    cap = 32
    while cap < TILED_BALANCE_CAPMUL * mean_group:
        cap *= 2
    stats = torch.empty(8, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spmm_tiled_map_stats", ic, M, Kd, cap, ptr(ptr_), None, None, ptr(stats), s)    # statistics only: 4-8 B per row
    st = stats.tolist()                                   # (the one read-back of the decision)
    skew = st[6] / max(mean_group, 1.0)
    TILED_BALANCE_STATS.update(cap=cap, class_rows=st[:6], natural_max_group=st[6], skew=skew, balanced=False)
    if skew <= TILED_BALANCE_SKEW:
        return None
    keys = torch.empty(M, dtype=torch.int64, device=dev)
    rows = torch.empty(M, dtype=torch.int32, device=dev)
    _ffi.call("spamd_spmm_tiled_map_stats", ic, M, Kd, cap, ptr(ptr_), ptr(keys), ptr(rows), ptr(stats), s)      # + the sort's keys and payload, rows per class
    st = stats.tolist()
    TILED_BALANCE_STATS.update(class_rows=st[:6])
    counts = _harr64(st[:6])
    groups = int(_ffi.lib().spamd_spmm_tiled_map_groups(counts))
    nseg = groups * ntiles
    upper = -(-nnz // epb) + nseg
    if groups <= 0 or upper >= 2 ** 31:
        return None
    keys, rows = sort_key_value(keys, rows, max(Kd, 1))  # longest rows first (stable): neighbours in a workgroup are equally long
    rowmap = torch.empty(groups * rg + 16, dtype=torch.int32, device=dev)
    gload = torch.empty(groups + 1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spmm_tiled_map_build", ic, M, cap, ptr(ptr_), ptr(rows), counts, ptr(rowmap), ptr(gload), s)
    vstart = exclusive_scan(gload)
    blocks = torch.empty((upper + slack) * 16, dtype=torch.int32, device=dev)
    blk_off = torch.empty(groups * (ntiles + 1), dtype=torch.int32, device=dev)
    state = torch.empty(1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spmm_tiled_inspect_mapped", vc, ic, M, Kd, groups, ptr(vals), ptr(ind), ptr(ptr_), ptr(rowmap), ptr(vstart),
              ptr(state), ptr(blk_off), ptr(blocks), s)
    TILED_BALANCE_STATS.update(balanced=True, groups=groups)
    lay = TiledLayout(blocks, blk_off, dtype, nnz / epb / max(nseg, 1) + 0.5, group_ends=True, pending=state if defer_check else None,
                      rowmap=rowmap, groups=groups)
    if not defer_check and int(state[0]) != 0:
        return None          # unsorted column indices: the caller's key-sort recipe (natural layout)
    return lay


def csc_tiled_layout(a_data, a_indices, a_indptr, M, Kd, dtype=None):
    """The one-pass inspector for a CSC operand (`a_indices` = row indices, `a_indptr` = Kd + 1 column pointers): the same
    block stream as `csr_tiled_layout(..., defer_check=True)` of the CSR twin would give the executor, without building the
    twin (late round 4: the reference-default tall operand paid 4.3 ms of CSC -> CSR at config 2's size before its first
    product).  Returns None when the shape is outside the one-pass builder (more than `direct_max` tiles).  Rows that do
    not ascend inside a column are reported like unsorted columns are: the layout's `pending` word, read behind the first
    product (`UnsortedColumns`; the lists are then empty and the caller converts to CSR)."""
    dev = require_hip(a_data, a_indices, a_indptr)
    if dtype is None:
        dtype = a_data.dtype if a_data.dtype in TILED_DTYPES else torch.float32
    dtype = torch_dtype(dtype)
    rg, kb, gpb, epb, slack, direct_max, _ = tiled_params(dtype)
    nnz = int(a_data.numel())
    ntiles = -(-Kd // kb)
    groups = -(-(-(-M // rg)) // gpb) * gpb
    nseg = groups * ntiles
    upper = -(-nnz // epb) + nseg
    if ntiles > direct_max or upper >= 2 ** 31 or groups == 0 or not TILED_ONE_PASS_INSPECTOR:
        return None
    if not index_dtype_ok(a_indices) or a_indices.dtype != a_indptr.dtype:
        a_indices, a_indptr = a_indices.to(torch.int64), a_indptr.to(torch.int64)
    vals = a_data.to(dtype).contiguous()
    if dtype == torch.int32:
        vals = vals.view(torch.float32)
    blocks = torch.empty((upper + slack) * 16, dtype=torch.int32, device=dev)
    blk_off = torch.empty(groups * (ntiles + 1), dtype=torch.int32, device=dev)
    split = torch.empty(int(_ffi.lib().spamd_spmm_tiled_inspect_csc_ws(M, Kd)), dtype=torch.int32, device=dev)   # (workspace)
    state = torch.empty(1, dtype=torch.int64, device=dev)
    _ffi.call("spamd_spmm_tiled_inspect_csc", code_of(_tiled_layout_dtype(dtype)), code_of(a_indices.dtype), M, Kd, ptr(vals),
              ptr(a_indices.contiguous()), ptr(a_indptr.contiguous()), ptr(split), ptr(state), ptr(blk_off), ptr(blocks),
              stream_ptr(dev))
    return TiledLayout(blocks, blk_off, dtype, nnz / epb / max(nseg, 1) + 0.5, group_ends=True, pending=state)


def dot_csr_ndarray_tiled(layout, out_shape, Kd, b, out=None, exact=False):
    """Executor: C = A @ B from the tiled layout; `exact` = separate multiply and add (the reference's arithmetic) instead
    of one FMA per term.  B provides whole column panels (float32: 128 columns, float64: 64; `b.shape[1]` a multiple of
    that, zero-padded by the caller); `out_shape[1]` may be narrower than B (any width since late round 4): the last panel
    then stores its leading columns only - no padded result, no slice afterwards."""
    blocks, blk_off, dtype = layout
    M, N = int(out_shape[0]), int(b.shape[1])
    Nout = int(out_shape[1])
    panel = 64 if dtype == torch.float64 else 128
    if N % panel or not (N - panel < Nout <= N):
        raise ValueError(f"tiled executor: B has {N} columns (whole {panel}-column panels expected), result {Nout}")
    last_cols = (Nout - (N - panel)) % panel    # 0 = the whole last panel
    dev = require_hip(blocks, blk_off, b)
    if b.dtype != dtype:
        raise TypeError(f"tiled layout holds {dtype} values, B is {b.dtype}")
    b = b.contiguous()
    if out is None:
        out = torch.empty((M, Nout), dtype=dtype, device=dev)
    # prefetch hint: ~1.5 x the mean number of 64-byte blocks per (row group, tile) list
    lists = max(int(blk_off.numel()) - 1, 1)
    mean_blocks = getattr(layout, "mean_blocks", None) or (int(blocks.numel()) // 16) / lists
    hint = min(64, max(6, int(1.5 * mean_blocks) + 3))
    if _TOUCH_OVERRIDE:
        hint = _TOUCH_OVERRIDE
    ends = _ffi.TILED_GROUP_ENDS if getattr(layout, "group_ends", False) else 0
    ints = _ffi.TILED_INT32 if dtype == torch.int32 else 0
    flags = (_ffi.EXACT_MULADD if exact and not ints else 0) | ends | ints | (hint << 8) | (last_cols << 16)
    rowmap = getattr(layout, "rowmap", None)
    if rowmap is not None:
        _ffi.call("spamd_spmm_tiled_mapped", code_of(_tiled_layout_dtype(dtype)), M, int(layout.groups), Kd, N, ptr(blocks), ptr(blk_off),
                  ptr(rowmap), ptr(b), N, ptr(out), Nout, flags, stream_ptr(dev))
    else:
        _ffi.call("spamd_spmm_tiled", code_of(_tiled_layout_dtype(dtype)), M, Kd, N, ptr(blocks), ptr(blk_off), ptr(b), N, ptr(out), Nout,
                  flags, stream_ptr(dev))
    pending = getattr(layout, "pending", None)
    if pending is not None:   # first product of a layout built with defer_check: the verdict is read now, behind the launch
        layout.pending = None
        words = pending.tolist()
        mean = getattr(layout, "skew_mean", None)
        if mean and len(words) >= 8 and words[7] > TILED_BALANCE_SKEW * max(mean, 1.0):
            layout.rebalance = True       # (the product just enqueued is correct; the owner rebuilds a balanced layout for the next ones)
        if int(words[0]) != 0:
            raise UnsortedColumns("tiled layout built from rows with unsorted column indices")
    return out
