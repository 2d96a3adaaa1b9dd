"""sparse_amd — MI355X-native (gfx950) backend for the pydata/sparse hot path.

Drop-in for the `sparse.numba_backend` names on that path: `COO`, `GCXS`, `tensordot`,
`matmul`, `dot`, `elemwise`, reductions and the NumPy protocols.  Arrays live in HBM as
PyTorch-ROCm tensors; all arithmetic is done by hand-written HIP kernels behind the C ABI of
`libsparse_amd.so` (include/sparse_amd.h).  There is no CPU fallback.
"""
from numpy import (  # noqa: F401  (the reference re-exports NumPy's ufuncs, __init__.py:1-83)
    abs, add, bitwise_and, bitwise_or, bitwise_xor, ceil, cos, cosh, divide, equal, exp, expm1, floor, greater,
    greater_equal, isfinite, isinf, isnan, less, less_equal, log, log1p, log2, log10, logical_and, logical_not,
    logical_or, logical_xor, maximum, minimum, multiply, negative, not_equal, positive, sign, sin, sinh, sqrt, square,
    subtract, tan, tanh, trunc,
)

from ._sparse_array import SparseArray
from ._coo import COO, as_coo
from ._gcxs import GCXS
from ._dot import dot, flush_warnings, matmul, tensordot
from ._umath import elemwise
from ._einsum import einsum
from ._batched import concatenate, stack
from ._broadcast import broadcast_to
from ._io import load_npz, save_npz
from ._api import (all, any, argwhere, asarray, astype, empty, empty_like, expand_dims, eye, full, full_like,
                   matrix_transpose, max, mean, min, moveaxis, nanmax, nanmean, nanmin, nanprod, nanreduce, nansum, nonzero,
                   ones, ones_like, permute_dims, prod, random, reshape, sddmm, squeeze, std, sum, var, vecdot, where, zeros,
                   zeros_like)
from ._ffi import HipBackendError

__all__ = ["COO", "GCXS", "SparseArray", "HipBackendError", "all", "any", "argwhere", "as_coo", "asarray", "astype", "broadcast_to",
           "concatenate", "dot", "einsum", "elemwise", "empty", "empty_like", "expand_dims", "eye", "full", "full_like", "load_npz", "matmul",
           "matrix_transpose", "max", "mean", "min", "moveaxis", "nanmax", "nanmean", "nanmin", "nanprod", "nanreduce", "nansum",
           "nonzero", "ones", "ones_like", "permute_dims", "prod", "random", "reshape", "save_npz", "sddmm", "squeeze", "stack", "std",
           "sum", "tensordot", "var", "vecdot", "where", "zeros", "zeros_like"]
