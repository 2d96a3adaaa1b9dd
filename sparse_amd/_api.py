"""Function surface of the hip backend on the hot path (reference
sparse/numba_backend/__init__.py:179-350, the subset SURVEY.md §8b names): thin wrappers that
delegate to the containers, like the reference's own (`_common.py:2077-2633`)."""
import builtins

import numpy as np
import torch

from . import _device as dev
from . import _kernels as K
from ._coo import COO, as_coo
from ._gcxs import GCXS
from ._sparse_array import SparseArray


def asarray(obj, /, *, dtype=None, format="coo", copy=False, device=None):
    """Convert to a sparse array of this backend (reference `asarray`, _common.py:2077-2135)."""
    if format not in {"coo", "gcxs"}:
        raise ValueError(f"{format} format not supported.")
    if isinstance(obj, SparseArray):
        out = obj.asformat(format)
    elif hasattr(obj, "tocoo") and type(obj).__module__.startswith("scipy.sparse"):
        out = (GCXS.from_scipy_sparse(obj, device=device) if format == "gcxs" else COO.from_scipy_sparse(obj, device=device))
    elif np.isscalar(obj) or isinstance(obj, (np.ndarray, torch.Tensor, list, tuple)):
        arr = obj if isinstance(obj, torch.Tensor) else np.asarray(obj)
        out = COO.from_numpy(arr, device=device).asformat(format)
    else:
        raise ValueError(f"{type(obj)} not supported.")
    if dtype is not None and out.dtype != np.dtype(dtype):
        out = out.astype(dtype)
    return out


def sddmm(s, a, b=None, *, bt=None):
    """Sampled dense-dense matmul: `s * (a @ b)` evaluated only at the stored positions of the
    2-D sparse mask `s` (the reference's formulation, examples/sddmm_example.py:51-52, forms the
    whole dense product first).  Pass `b` (K x N) or its transpose `bt` (N x K, K-contiguous:
    avoids a device transpose).  Dense operands may be bfloat16/float32/float64 torch tensors
    (bf16/fp32 accumulate in fp32) or float ndarrays.  Result has the format of `s`, zeros pruned."""
    from ._utils import check_zero_fill_value

    check_zero_fill_value(s)
    if s.ndim != 2:
        raise ValueError("sddmm needs a 2-D sparse mask")
    out_gcxs = isinstance(s, GCXS)
    sc = s if isinstance(s, COO) else s.tocoo()
    at = dev.to_device(a, sc.device)
    if (b is None) == (bt is None):
        raise ValueError("pass exactly one of b / bt")
    btt = dev.to_device(bt, sc.device) if bt is not None else dev.to_device(b, sc.device).t().contiguous()
    if at.shape[0] != s.shape[0] or btt.shape[0] != s.shape[1] or at.shape[1] != btt.shape[1]:
        raise ValueError("shape-mismatch for sum")
    vals = K.sddmm_coo(sc.coords, sc.data, at, btt)
    out = COO(sc.coords, vals, shape=s.shape, has_duplicates=False, sorted=True, prune=True)
    return out.asformat("gcxs", compressed_axes=s.compressed_axes) if out_gcxs else out


def random(shape, density=None, nnz=None, random_state=None, format="coo", fill_value=None, idx_dtype=None,
           dtype=np.float64, device=None, **kwargs):
    """Random sparse array generated ON THE DEVICE: uniform-without-replacement positions,
    values U[0, 1) — the distribution of the reference's `sparse.random` (_utils.py:221-346; its
    Vitter sampler is a sequential loop).  The random stream is torch's, not NumPy's: same
    distribution, different numbers."""
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
    size = 1
    for s in shape:
        size *= s
    if density is not None and nnz is not None:
        raise ValueError("Specify either density or nnz, not both")
    if density is None and nnz is None:
        density = 0.01
    if density is not None and not 0 <= density <= 1:
        raise ValueError(f"density {density} is not in the unit interval")
    if nnz is None:
        nnz = int(size * density)
    if not 0 <= nnz <= size:
        raise ValueError(f"Cannot generate {nnz} nonzero elements for an array with {size} total elements.")
    d = torch.device(device) if device is not None else dev.default_device()
    g = torch.Generator(device=d)
    g.manual_seed(int(random_state) if random_state is not None else torch.seed() % (2 ** 31))
    if nnz > size // 2 and size <= 2 ** 31:
        keys = torch.randperm(size, generator=g, device=d)[:nnz].sort().values
    else:
        keys = torch.empty(0, dtype=torch.int64, device=d)
        while keys.numel() < nnz:
            need = nnz - keys.numel()
            draw = torch.randint(0, builtins.max(size, 1), (int(need * 1.05) + 1024,), generator=g, device=d)
            keys = torch.unique(torch.cat([keys, draw]))
        if keys.numel() > nnz:
            keep = torch.ones(keys.numel(), dtype=torch.bool, device=d)
            keep[torch.randperm(keys.numel(), generator=g, device=d)[: keys.numel() - nnz]] = False
            keys = keys[keep]
    data = torch.rand(nnz, generator=g, device=d, dtype=torch.float64).to(dev.torch_dtype(dtype))
    it = torch.int64 if idx_dtype is None or np.dtype(idx_dtype).itemsize > 4 else torch.int32
    coords = K.delinearize(keys.contiguous(), shape, it)
    out = COO(coords, data, shape=shape, has_duplicates=False, sorted=True, fill_value=fill_value)
    out._keys = keys
    return out.asformat(format, **kwargs) if format != "coo" else out


def _reduction(name):
    def f(x, /, *, axis=None, keepdims=False, **kw):
        return getattr(x, name)(axis=axis, keepdims=keepdims, **kw)

    f.__name__ = name
    return f


sum = _reduction("sum")
prod = _reduction("prod")
max = _reduction("max")
min = _reduction("min")
mean = _reduction("mean")
any = _reduction("any")
all = _reduction("all")
var = _reduction("var")
std = _reduction("std")


def astype(x, dtype, /, *, copy=True):
    return x.astype(dtype, copy=copy)


def reshape(x, /, shape, *, copy=None):
    return x.reshape(shape)


def permute_dims(x, /, axes=None):
    return x.transpose(axes)


def matrix_transpose(x, /):
    return x.mT


def vecdot(x1, x2, /, *, axis=-1):
    """sum(x1 * x2, axis) (reference `vecdot`, _common.py)."""
    return (x1 * x2).sum(axis=axis)
