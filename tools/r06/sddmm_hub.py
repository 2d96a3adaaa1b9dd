"""SDDMM with a hub row / hub column in the mask: ms per call against the uniform mask (10^7 stored elements, K = 64)."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time

g = torch.Generator(device="cuda").manual_seed(3)
M, N, nnz, Kd = 1_000_000, 1_000_000, 10_000_000, 64
base = torch.randint(0, M * N, (nnz,), device="cuda", generator=g)
hubrow = torch.randperm(N, device="cuda", generator=g)[:500_000] + 77 * N
hubcol = torch.randperm(M, device="cuda", generator=g)[:500_000] * N + 99
x = torch.rand(M, Kd, device="cuda")
yt = torch.rand(N, Kd, device="cuda")
for label, lin in (("uniform", base), ("+row of 5e5", torch.cat([base, hubrow])), ("+column of 5e5", torch.cat([base, hubcol])),
                   ("+both", torch.cat([base, hubrow, hubcol]))):
    lin = torch.unique(lin)
    vals = torch.rand(lin.numel(), device="cuda") + 0.1
    c = sp.COO._from_sorted_keys(lin, vals, (M, N), 0.0, torch.int64)
    row = [label]
    for dt in (torch.float32, torch.bfloat16):
        f = lambda: sp.sddmm(c, x.to(dt), bt=yt.to(dt))
        xa, ya = x.to(dt), yt.to(dt)
        f = lambda: sp.sddmm(c, xa, bt=ya)
        f()
        f()
        row.append(f"{str(dt)[6:]} {dev_time(f, 5):.2f} ms")
    print(" | ".join(row), flush=True)
