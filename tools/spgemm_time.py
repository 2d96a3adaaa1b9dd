import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
g = sp.random((n4, n4), density=1e-3, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
def run(local):
    K.SPGEMM_ROW_LOCAL = local
    torch.cuda.empty_cache()
    for _ in range(2): c = g @ g
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): c = g @ g
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 3 * 1e3, c
tl, cl = run(True)
tg, cg = run(False)
print("row-local stats:", K.SPGEMM_STATS)
print(f"n={n4} nnz={g.nnz}: row-local {tl:.2f} ms, global ESC {tg:.2f} ms, out nnz {cl.nnz}")
print("identical:", torch.equal(cl.indptr.long(), cg.indptr.long()), torch.equal(cl.indices.long(), cg.indices.long()), torch.equal(cl.data, cg.data))
