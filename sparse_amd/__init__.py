"""sparse_amd — MI355X-native (gfx950) backend for the pydata/sparse hot path.

Drop-in for the `sparse.numba_backend` names on that path: `COO`, `GCXS`, `tensordot`,
`matmul`, `dot`, `elemwise`, reductions and the NumPy protocols.  Arrays live in HBM as
PyTorch-ROCm tensors; all arithmetic is done by hand-written HIP kernels behind the C ABI of
`libsparse_amd.so` (include/sparse_amd.h).  There is no CPU fallback.
"""
from ._sparse_array import SparseArray
from ._coo import COO, as_coo
from ._gcxs import GCXS
from ._dot import dot, matmul, tensordot
from ._ffi import HipBackendError

__all__ = ["COO", "GCXS", "SparseArray", "as_coo", "dot", "matmul", "tensordot", "HipBackendError"]
