"""Sparse x sparse with a hub row / hub column in the operands: ms per product against the uniform matrix."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time

g = torch.Generator(device="cuda").manual_seed(3)
n, per = 100_000, 10
base = torch.randint(0, n * n, (n * per,), device="cuda", generator=g)
hubrow = torch.randperm(n, device="cuda", generator=g)[:50_000] + 77 * n
hubcol = torch.randperm(n, device="cuda", generator=g)[:50_000] * n + 99
for label, lin in (("uniform", base), ("+row of 5e4", torch.cat([base, hubrow])), ("+column of 5e4", torch.cat([base, hubcol])),
                   ("+both", torch.cat([base, hubrow, hubcol]))):
    lin = torch.unique(lin)
    for dt in (torch.float32, torch.float64):
        vals = torch.rand(lin.numel(), device="cuda", dtype=dt) + 0.1
        a = sp.GCXS(sp.COO._from_sorted_keys(lin, vals, (n, n), 0.0, torch.int64), compressed_axes=(0,))
        row = [label, str(dt)[6:]]
        for name, f in (("a@a", lambda: a @ a), ("a@a.T", lambda: a @ a.T), ("a.T@a", lambda: a.T @ a)):
            try:
                r = f()
                f()
                row.append(f"{name} {dev_time(f, 3):.2f} ms (nnz {r.nnz})")
            except Exception as e:
                row.append(f"{name} {type(e).__name__}:{str(e)[:40]}")
        print(" | ".join(row), flush=True)
