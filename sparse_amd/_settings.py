"""Environment flags, read once at import — same names as the reference
(sparse/numba_backend/_settings.py:5-6)."""
import os

AUTO_DENSIFY = bool(int(os.environ.get("SPARSE_AUTO_DENSIFY", "0")))
WARN_ON_TOO_DENSE = bool(int(os.environ.get("SPARSE_WARN_ON_TOO_DENSE", "0")))
# hip-backend extras
NAN_CHECK = bool(int(os.environ.get("SPARSE_AMD_NAN_CHECK", "1")))  # matmul's NaN RuntimeWarning pass
# "sync": the NaN warning is raised before matmul returns (the reference's behaviour: the host waits for the scan kernels,
# never for the product).  "deferred": the scans run all the same, but their verdicts are polled without blocking - the
# warning is raised by a later sparse_amd.matmul call (or by sparse_amd.flush_warnings()); for loops of short products
# whose launch rate the per-product host wait would otherwise bound (multi-GPU strong scaling: 0.12 ms kernels)
NAN_WARNING = os.environ.get("SPARSE_AMD_NAN_WARNING", "sync")
EXACT_MULADD = bool(int(os.environ.get("SPARSE_AMD_EXACT", "0")))  # bit-exact mul+add instead of FMA
# CSR x dense products: "auto" builds the K-tiled block stream of a matrix (csrc/spmm_tiled.hip) at its first
# eligible product (fp32, N % 128 == 0, FMA mode, large enough) and caches it on the array; "never" keeps the
# row-group kernel
TILED_SPMM = os.environ.get("SPARSE_AMD_TILED_SPMM", "auto")


class ArrayNamespaceInfo:
    """Array-API inspection object (reference `_settings.py:24-46`): NumPy's dtypes; the devices are the HIP devices the
    arrays live on, not "cpu"."""

    def __init__(self):
        import numpy as np

        self.np_info = np.__array_namespace_info__()

    def capabilities(self):
        return {"boolean indexing": False, "data-dependent shapes": True, "max dimensions": 16}      # MAX_NDIM of the C ABI

    def default_device(self):
        from ._device import default_device

        return default_device()

    def default_dtypes(self, *, device=None):
        return self.np_info.default_dtypes(device=None)

    def devices(self):
        import torch

        return tuple(torch.device("cuda", i) for i in range(torch.cuda.device_count()))

    def dtypes(self, *, device=None, kind=None):
        return self.np_info.dtypes(device=None, kind=kind)


def __array_namespace_info__():
    return ArrayNamespaceInfo()
