"""sum(axis=0) of 3-D COO arrays with 10^6 stored elements and S = 16 .. 2000 leading indices: slab merge against key sort."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
for S, n in ((16, 10 ** 6), (64, 10 ** 6), (100, 10 ** 6), (300, 10 ** 6), (1000, 10 ** 6), (2000, 10 ** 6), (1000, 2 * 10 ** 6), (100, 4 * 10 ** 6), (1000, 10 ** 5)):
    shape = (S, 1000, 1000)
    g = torch.Generator(device="cuda").manual_seed(S)
    lin = torch.unique(torch.randint(0, S * 10 ** 6, (int(n * 1.1),), device="cuda", generator=g, dtype=torch.int64))[:n]
    x = sp.COO._from_sorted_keys(lin, torch.rand(lin.numel(), device="cuda", generator=g, dtype=torch.float64), shape, np.float64(0), torch.int64)
    out = []
    res = {}
    for flag in (True, False):
        K.LEAD_LAST = flag
        K.LEAD_LAST_STATS.clear()
        for _ in range(3): r = x.sum(axis=0)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): r = x.sum(axis=0)
        torch.cuda.synchronize()
        res[flag] = r
        out.append(f"{'merge' if flag else 'sort'} {(time.perf_counter() - t) / 20 * 1e3:.3f} ms{' (' + str(K.LEAD_LAST_STATS.get('ranges')) + ' ranges)' if flag else ''}")
    K.LEAD_LAST = True
    same = torch.equal(res[True].linear_loc(), res[False].linear_loc()) and torch.equal(res[True].data, res[False].data)
    print(f"S={S} nnz={x.nnz}: " + ", ".join(out) + f", identical {same}")
