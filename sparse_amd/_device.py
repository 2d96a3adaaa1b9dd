"""Device-memory plumbing: PyTorch-ROCm tensors hold every array in HBM.

torch is used here for allocation, streams and host<->device copies only; arithmetic on
the hot path goes through libsparse_amd.so (see _kernels.py).
"""
import numpy as np
import torch

from . import _ffi

_NP2CODE = {np.dtype("float32"): _ffi.F32, np.dtype("float64"): _ffi.F64,
            np.dtype("int32"): _ffi.I32, np.dtype("int64"): _ffi.I64}
_T2NP = {torch.float32: np.dtype("float32"), torch.float64: np.dtype("float64"),
         torch.int32: np.dtype("int32"), torch.int64: np.dtype("int64"),
         torch.int16: np.dtype("int16"), torch.int8: np.dtype("int8"),
         torch.uint8: np.dtype("uint8"), torch.bool: np.dtype("bool"),
         torch.float16: np.dtype("float16"), torch.complex64: np.dtype("complex64"),
         torch.complex128: np.dtype("complex128")}
_NP2T = {v: k for k, v in _T2NP.items()}
_NP2T[np.dtype("uint64")] = torch.int64  # reference builds empty coords as uintp
_NP2T[np.dtype("uint32")] = torch.int64
_NP2T[np.dtype("uint16")] = torch.int32


def np_dtype(t):
    """NumPy dtype of a torch tensor / torch dtype / numpy-like dtype."""
    if isinstance(t, torch.Tensor):
        t = t.dtype
    if isinstance(t, torch.dtype):
        if t is torch.bfloat16:
            raise TypeError("bfloat16 has no NumPy dtype")
        return _T2NP[t]
    return np.dtype(t)


def torch_dtype(dt):
    if isinstance(dt, torch.dtype):
        return dt
    return _NP2T[np.dtype(dt)]


def code_of(dt):
    """C-ABI dtype code for a value/index dtype; TypeError if the HIP path lacks it."""
    if isinstance(dt, torch.dtype) and dt is torch.bfloat16:
        return _ffi.BF16
    d = np_dtype(dt)
    if d not in _NP2CODE:
        raise TypeError(f"dtype {d} is not supported by the hip backend (supported: "
                        "float32, float64, int32, int64)")
    return _NP2CODE[d]


def default_device():
    """The HIP device arrays are created on.  Raises if there is none: no CPU fallback."""
    if not torch.cuda.is_available():
        raise _ffi.HipBackendError("no HIP device visible: sparse_amd computes only on MI355X "
                                   "(there is deliberately no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def is_dense(x):
    return isinstance(x, (np.ndarray, torch.Tensor))


def to_device(x, device=None, dtype=None):
    """ndarray / tensor / sequence -> contiguous-ish tensor on `device` (default: HIP device)."""
    if device is None:
        device = default_device()
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.asarray(x)
        if a.dtype.kind == "u" and a.dtype.itemsize >= 2:
            a = a.astype(np.int64 if a.dtype.itemsize >= 4 else np.int32)
        if not a.flags.writeable or any(s < 0 for s in a.strides):
            a = np.array(a)
        t = torch.from_numpy(a) if a.ndim > 0 else torch.tensor(a.item(), dtype=torch_dtype(a.dtype))
    if dtype is not None:
        dtype = torch_dtype(dtype)
    return t.to(device=device, dtype=dtype, non_blocking=False)


def to_numpy(t):
    return t.detach().cpu().numpy()


def require_hip(*tensors):
    """All tensors on one HIP device; returns it.  Loud failure otherwise."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _ffi.HipBackendError(
                "sparse_amd kernels need device (HIP) tensors; got a CPU tensor and there is "
                "deliberately no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device):
    """raw hipStream_t of torch's current stream on `device` (the C-level query when torch offers it: every C-ABI call
    passes a stream, and the Python-level `torch.cuda.current_stream` costs ~5 us per call)"""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return 0 if t is None or t.numel() == 0 else t.data_ptr()
