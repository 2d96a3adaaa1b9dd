"""sum(axis=0) of COO (1000, 1000, 1000) with 10^8 stored elements: slab merge against key sort (ms per call)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
g = torch.Generator(device="cuda").manual_seed(1)
lin = torch.randint(0, 10 ** 9, (int(n * 1.06),), device="cuda", generator=g, dtype=torch.int64)
lin = torch.unique(lin)[:n]          # sorted, duplicate-free
x = sp.COO._from_sorted_keys(lin, torch.rand(lin.numel(), device="cuda", generator=g, dtype=torch.float64), (1000, 1000, 1000), np.float64(0), torch.int64)
def run(reps=5):
    for _ in range(2): r = x.sum(axis=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = x.sum(axis=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, r
res = {}
for flag in (True, False):
    K.LEAD_LAST = flag
    K.LEAD_LAST_STATS.clear()
    ms, r = run()
    res[flag] = r
    print(f"nnz {x.nnz}: merge={flag}: {ms:.3f} ms, result nnz {r.nnz}, stats {K.LEAD_LAST_STATS}")
K.LEAD_LAST = True
print("same keys:", torch.equal(res[True].linear_loc(), res[False].linear_loc()), "same values:", torch.equal(res[True].data, res[False].data))
