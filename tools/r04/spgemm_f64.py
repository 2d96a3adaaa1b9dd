"""config-5 share with the reference's default dtypes (float64 values, int64 indices): buckets vs bitmap (column ranges)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n, share = 1_000_000, 8
gB = sp.random((n, n), density=1e-4, random_state=7, dtype=np.float64, idx_dtype=np.int64, format="gcxs", compressed_axes=(0,))
rows = n // share
p1 = int(gB.indptr[rows])
gA = sp.GCXS((gB.data[:p1].contiguous(), gB.indices[:p1].contiguous(), gB.indptr[:rows + 1].contiguous()), shape=(rows, n), compressed_axes=(0,))
res = {}
for name, flag in (("buckets", False), ("bitmap", True)):
    K.SPGEMM_BITMAP = flag
    for _ in range(2): c = gA @ gB
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): c = gA @ gB
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 3 * 1e3:.2f} ms per product, nnz {c.nnz}, {K.SPGEMM_STATS}", flush=True)
    res[name] = (c.data.clone(), c.indices.clone(), c.indptr.clone())
    del c; torch.cuda.empty_cache()
print("bit-identical:", all(torch.equal(x, y) for x, y in zip(res["buckets"], res["bitmap"])))
