"""einsum patterns (reference _common.py:1163-1476) at 10^7 stored elements: ms per call."""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp

g = torch.Generator(device="cuda").manual_seed(1)
def coo(shape, nnz, seed):
    size = int(np.prod(shape))
    lin = torch.unique(torch.randint(0, size, (nnz,), device="cuda", generator=g))
    return sp.COO._from_sorted_keys(lin, torch.rand(lin.numel(), device="cuda", dtype=torch.float64) + 0.1, shape, 0.0, torch.int64)
x = coo((100_000, 10_000), 10_000_000, 1)
y = coo((100_000, 10_000), 10_000_000, 2)
z = coo((10_000, 5000), 1_000_000, 3)
t3 = coo((100, 1000, 10_000), 10_000_000, 4)
v = torch.rand(10_000, device="cuda", dtype=torch.float64)
d = torch.rand(10_000, 16, device="cuda", dtype=torch.float64)
cases = [("ij,ij->i", (x, y)), ("ij,ij->ij", (x, y)), ("ij,ij->", (x, y)), ("ij->j", (x,)), ("ij->ji", (x,)), ("ij,jk->ik (sp,dense)", (x, d)),
         ("ij,j->i (sp,dense)", (x, v)), ("ij,jk->ik (sp,sp)", (x, z)), ("ijk,k->ij", (t3, v)), ("ijk,ik->ij?", None), ("ii->i", (coo((10_000, 10_000), 1_000_000, 5),)),
         ("ijk->kji", (t3,)), ("ijk,jk->i", (t3, coo((1000, 10_000), 1_000_000, 6)))]
for name, ops in cases:
    if ops is None:
        continue
    sub = name.split(" ")[0]
    f = lambda: sp.einsum(sub, *ops)
    try:
        r = f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        print(f"{name:26s} {(time.perf_counter() - t0) / 3 * 1e3:9.2f} ms  -> {type(r).__name__} {getattr(r, 'shape', None)} nnz {getattr(r, 'nnz', None)}", flush=True)
    except Exception as e:
        print(f"{name:26s} {type(e).__name__}: {str(e)[:100]}", flush=True)
print({k: v for k, v in sp.fallback_stats().items() if k != "recent"})
