"""G @ G over sizes and row lengths: ms per product, 10^9 products per second, the kernel family that ran"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _kernels as K
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
for dtype, idt in ((np.float32, np.int32), (np.float64, np.int64)):
    for n in (1000, 10_000, 100_000, 1_000_000):
        for per_row in (2, 10, 30, 100):
            if n * per_row * per_row > 3e9 or per_row >= n:
                continue
            g = sp.random((n, n), density=per_row / n, random_state=7, dtype=dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
            try:
                K.SPGEMM_STATS.clear() if hasattr(K, "SPGEMM_STATS") and hasattr(K.SPGEMM_STATS, "clear") else None
                ms, c = t(lambda: g @ g)
                prods = g.nnz * per_row
                print(f"{np.dtype(dtype).name:8s} n={n:8d} nnz/row={per_row:4d} nnz={g.nnz:10d}: {ms:9.3f} ms  {prods / ms / 1e6:7.2f} Gprod/s  out nnz {c.nnz:11d}  {getattr(K, 'SPGEMM_STATS', '')}", flush=True)
            except Exception as e:      # noqa: BLE001
                print(f"n={n} per_row={per_row}: {type(e).__name__} {str(e)[:100]}", flush=True)
            del g
            torch.cuda.empty_cache()
