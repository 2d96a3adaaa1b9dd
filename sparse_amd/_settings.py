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
