"""sparse_amd — MI355X-native (gfx950) backend for the pydata/sparse hot path.

Drop-in for the `sparse.numba_backend` names on that path: `COO`, `GCXS`, `tensordot`,
`matmul`, `dot`, `elemwise`, reductions and the NumPy protocols.  Arrays live in HBM as
PyTorch-ROCm tensors; all arithmetic is done by hand-written HIP kernels behind the C ABI of
`libsparse_amd.so` (include/sparse_amd.h).  There is no CPU fallback.
"""
from numpy import (  # noqa: F401  (the reference re-exports NumPy's ufuncs, dtypes and constants, __init__.py:1-83)
    add, bitwise_and, bitwise_not, bitwise_or, bitwise_xor, ceil, complex64, complex128, conj, copysign, cos, cosh, divide, e,
    exp, expm1, finfo, float16, float32, float64, floor, floor_divide, greater, greater_equal, hypot, iinfo, inf, int8, int16,
    int32, int64, isfinite, less, less_equal, log, log1p, log2, log10, logaddexp, logical_and, logical_not, logical_or,
    logical_xor, maximum, minimum, multiply, nan, negative, newaxis, nextafter, not_equal, pi, positive, reciprocal, remainder,
    sign, signbit, sin, sinh, sqrt, square, subtract, tan, tanh, trunc, uint8, uint16, uint32, uint64,
)
from numpy import arccos as acos  # noqa: F401  (array-API spellings, __init__.py:71-83)
from numpy import arccosh as acosh  # noqa: F401
from numpy import arcsin as asin  # noqa: F401
from numpy import arcsinh as asinh  # noqa: F401
from numpy import arctan as atan  # noqa: F401
from numpy import arctan2 as atan2  # noqa: F401
from numpy import arctanh as atanh  # noqa: F401
from numpy import bool_ as bool  # noqa: F401, A004
from numpy import invert as bitwise_invert  # noqa: F401
from numpy import isdtype  # noqa: F401
from numpy import left_shift as bitwise_left_shift  # noqa: F401
from numpy import power as pow  # noqa: F401, A004
from numpy import right_shift as bitwise_right_shift  # noqa: F401

from ._sparse_array import SparseArray
from ._coo import COO, as_coo
from ._gcxs import GCXS
from ._dot import dot, flush_warnings, matmul, tensordot
from ._umath import elemwise, fallback_stats
from ._einsum import einsum
from ._batched import concatenate, stack
from ._broadcast import broadcast_to
from ._io import load_npz, save_npz
from ._api import (all, any, argwhere, asarray, astype, empty, empty_like, expand_dims, eye, full, full_like,
                   matrix_transpose, max, mean, min, moveaxis, nanmax, nanmean, nanmin, nanprod, nanreduce, nansum, nonzero,
                   ones, ones_like, permute_dims, prod, random, reshape, sddmm, squeeze, std, sum, var, vecdot, where, zeros,
                   zeros_like)
from ._array_api import (abs, argmax, argmin, asCOO, asnumpy, broadcast_arrays, broadcast_shapes, can_cast, clip, concat, diagonal, diagonalize,
                         equal, flip, imag, isinf, isnan, isneginf, isposinf, kron, outer, pad, real,
                         result_type, roll, round, sort, tril, triu)
from ._ffi import HipBackendError
from ._settings import __array_namespace_info__  # noqa: F401

__array_api_version__ = "2025.12"   # as the reference declares (sparse/__init__.py:7)

__all__ = ["COO", "GCXS", "SparseArray", "HipBackendError", "abs", "acos", "acosh", "add", "all", "any", "argmax", "argmin", "argwhere", "asCOO", "as_coo",
           "asarray", "asin", "asinh", "asnumpy", "astype", "atan", "atan2", "atanh", "bitwise_and", "bitwise_invert",
           "bitwise_left_shift", "bitwise_not", "bitwise_or", "bitwise_right_shift", "bitwise_xor", "bool", "broadcast_arrays",
           "broadcast_shapes", "broadcast_to", "can_cast", "ceil", "clip", "complex128", "complex64", "concat", "conj", "copysign",
           "cos", "cosh", "diagonal", "diagonalize", "divide", "e", "equal", "exp", "expm1", "finfo", "flip", "float16",
           "float32", "float64", "floor", "floor_divide", "greater", "greater_equal", "hypot", "iinfo", "imag", "inf", "int16",
           "int32", "int64", "int8", "isdtype", "isfinite", "isinf", "isnan", "isneginf", "isposinf", "kron", "less", "less_equal",
           "log", "log10", "log1p", "log2", "logaddexp", "logical_and", "logical_not", "logical_or", "logical_xor", "maximum",
           "minimum", "multiply", "nan", "negative", "newaxis", "nextafter", "not_equal", "outer", "pad", "pi", "positive", "pow",
           "real", "reciprocal", "remainder", "result_type", "roll", "round", "sign", "signbit", "sin", "sinh", "sqrt",
           "square", "subtract", "tan", "tanh", "sort", "tril", "triu", "trunc", "uint16", "uint32", "uint64", "uint8", 
           "concatenate", "dot", "einsum", "elemwise", "empty", "empty_like", "expand_dims", "eye", "full", "full_like", "load_npz", "matmul",
           "matrix_transpose", "max", "mean", "min", "moveaxis", "nanmax", "nanmean", "nanmin", "nanprod", "nanreduce", "nansum",
           "nonzero", "ones", "ones_like", "permute_dims", "prod", "random", "reshape", "save_npz", "sddmm", "squeeze", "stack", "std",
           "sum", "tensordot", "var", "vecdot", "where", "zeros", "zeros_like"]
