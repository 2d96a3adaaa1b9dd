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
K.SPGEMM_SMALL_SECOND = False
for dtype, idt, n, per_row in ((np.float32, np.int32, 1000, 100), (np.float32, np.int32, 10_000, 100), (np.float64, np.int64, 10_000, 100), (np.float64, np.int64, 20_000, 100),
                               (np.float64, np.int64, 30_000, 100), (np.float64, np.int64, 20_000, 140), (np.float32, np.int32, 20_000, 140), (np.float32, np.int32, 30_000, 170),
                               (np.float64, np.int64, 3000, 60), (np.float32, np.int32, 25_000, 110)):
    g = sp.random((n, n), density=per_row / n, random_state=7, dtype=dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
    ms, c = t(lambda: g @ g)
    print(f"{np.dtype(dtype).name} n={n} nnz/row={per_row}: {ms:.3f} ms  {g.nnz * per_row / ms / 1e6:.2f} Gprod/s  declined {K.SPGEMM_STATS.get('heavy_or_declined')} of {n}  max_prod {K.SPGEMM_STATS.get('max_prod')}", flush=True)
