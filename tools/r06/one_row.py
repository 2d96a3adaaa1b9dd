"""A matrix of ONE row (a sparse vector as a 1 x K matrix) times dense: ms per product by stored elements and width."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time
from sparse_amd import _dot as D

g = torch.Generator(device="cuda").manual_seed(3)
Kd = 2_000_000
for nnz in (50_000, 200_000, 1_000_000):
    lin = torch.unique(torch.randint(0, Kd, (nnz,), device="cuda", generator=g))
    vals = torch.rand(lin.numel(), device="cuda") + 0.1
    row = [f"1 x {Kd}, {lin.numel()} stored"]
    for n in (1, 16, 128):
        b = torch.rand(Kd, n, device="cuda")
        ts = []
        for on in (False, True):
            D.HOT_ROW_SPLIT = on
            a = sp.GCXS(sp.COO._from_sorted_keys(lin, vals, (1, Kd), 0.0, torch.int64), compressed_axes=(0,))
            a @ b
            a @ b
            ts.append(dev_time(lambda: a @ b, 3))
        row.append(f"N={n}: {ts[0]:.2f} -> {ts[1]:.2f}")
    D.HOT_ROW_SPLIT = True
    print(" | ".join(row), flush=True)
