"""One of 8 row blocks of config 5 (GCXS 125000 x 1e6 @ GCXS 1e6 x 1e6, 1e-4): time per product."""
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
for _ in range(1): c = gA @ gB
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): c = gA @ gB
torch.cuda.synchronize()
print(f"share: {(time.perf_counter() - t) / reps * 1e3:.2f} ms per product, out nnz {c.nnz}, stats {K.SPGEMM_STATS}")
