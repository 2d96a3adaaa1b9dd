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
for dtype, idt, n, per_row in ((np.float32, np.int32, 1000, 100), (np.float64, np.int64, 1000, 100), (np.float32, np.int32, 10_000, 100), (np.float64, np.int64, 10_000, 100),
                               (np.float64, np.int32, 10_000, 100), (np.float32, np.int32, 100_000, 100), (np.float64, np.int64, 100_000, 100), (np.float64, np.int64, 100_000, 70),
                               (np.float64, np.int64, 30_000, 100), (np.float32, np.int32, 30_000, 100), (np.float32, np.int32, 3000, 60), (np.float64, np.int64, 3000, 60)):
    g = sp.random((n, n), density=per_row / n, random_state=7, dtype=dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
    ms, c = t(lambda: g @ g)
    print(f"{np.dtype(dtype).name} {np.dtype(idt).name} n={n} nnz/row={per_row}: {ms:.3f} ms  {g.nnz * per_row / ms / 1e6:.2f} Gprod/s\n     {K.SPGEMM_STATS}", flush=True)
