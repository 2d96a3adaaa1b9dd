"""Raw-array kernel layer: the analogue of the reference's jitted `_dot_*` kernels.

Each function takes device tensors (raw arrays, no containers), allocates and returns fresh
outputs — the reference's ownership rule — and launches exactly one C-ABI
entry point of libsparse_amd.so on the current HIP stream.  No arithmetic happens in torch.
"""
import torch

from . import _ffi
from ._device import code_of, np_dtype, ptr, require_hip, stream_ptr, torch_dtype
import numpy as np


def dot_dtype(dt1, dt2):
    """Result dtype rule of every `_dot` kernel: `(zeros(dt1) * zeros(dt2)).dtype`
    (reference sparse/numba_backend/_common.py:635-636)."""
    return (np.zeros((), dtype=np_dtype(dt1)) * np.zeros((), dtype=np_dtype(dt2))).dtype


def _unify_index(*idx):
    """Index arrays handed to one kernel share a width: int32 only if all are int32."""
    want = torch.int32 if all(i.dtype == torch.int32 for i in idx) else torch.int64
    return [i if i.dtype == want else i.to(want) for i in idx], want


def dot_csr_ndarray(out_shape, a_data, a_indices, a_indptr, b, *, exact=False, out=None):
    """C = A @ B, A in CSR, B dense row-major, C dense — `_dot_csr_ndarray`
    (reference _common.py:720-755).  `exact=True` reproduces the reference's separate
    multiply/add bit-for-bit; the default uses one FMA per term (same k-ascending order)."""
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
    _ffi.call("spamd_spmm_csr", vcode, code_of(it), M, K, N, ptr(a_data), ptr(a_indices),
              ptr(a_indptr), ptr(b), max(N, 1), ptr(out), max(N, 1),
              _ffi.EXACT_MULADD if exact else 0, stream_ptr(dev))
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
