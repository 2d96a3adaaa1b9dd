"""bucket kernels vs bitmap kernel by products per row (n_col = 1e6, 60000 output rows)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n = 1_000_000
for per in (20, 30, 36, 40, 45, 64, 80, 100):
    dens = per / n
    gB = sp.random((n, n), density=dens, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    rows = 60_000
    p1 = int(gB.indptr[rows])
    gA = sp.GCXS((gB.data[:p1].contiguous(), gB.indices[:p1].contiguous(), gB.indptr[:rows + 1].contiguous()), shape=(rows, n), compressed_axes=(0,))
    out = []
    for flag in (False, True):
        K.SPGEMM_BITMAP, K.SPGEMM_BITMAP_MIN_MEAN = flag, 0
        for _ in range(2): c = gA @ gB
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): c = gA @ gB
        torch.cuda.synchronize()
        out.append(((time.perf_counter() - t) / 3 * 1e3, K.SPGEMM_STATS.get("kernel")))
        del c
    print(f"{per}x{per} = {per*per} products/row: buckets {out[0][0]:.2f} ms, bitmap {out[1][0]:.2f} ms ({out[1][1]})", flush=True)
    del gA, gB
    torch.cuda.empty_cache()
