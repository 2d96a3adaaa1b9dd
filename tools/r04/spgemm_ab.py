"""config-5 share (one of 8 row blocks): the bucket kernels against the bitmap kernel, ms per product through `a @ b`
and for the raw kernel layer; results compared bit for bit."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n, share = 1_000_000, 8
gB = sp.random((n, n), density=1e-4, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
rows = n // share
p1 = int(gB.indptr[rows])
gA = sp.GCXS((gB.data[:p1].contiguous(), gB.indices[:p1].contiguous(), gB.indptr[:rows + 1].contiguous()), shape=(rows, n), compressed_axes=(0,))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for name, flag in (("buckets", False), ("bitmap", True)):
    K.SPGEMM_BITMAP = flag
    for _ in range(2):
        c = gA @ gB
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        c = gA @ gB
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"{name}: {ms:.2f} ms per product, out nnz {c.nnz}, stats {K.SPGEMM_STATS}", flush=True)
    res[name] = (c.data.clone(), c.indices.clone(), c.indptr.clone())
    del c
    torch.cuda.empty_cache()
same = all(torch.equal(x, y) for x, y in zip(res["buckets"], res["bitmap"]))
print("bit-identical:", same)
