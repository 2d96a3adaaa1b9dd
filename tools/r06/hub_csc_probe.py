"""Hub rows in a CSC-stored operand (the reference's default layout for a tall matrix): steady-state product, ms, beside the
CSR-stored twin of the same matrix."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time

g = torch.Generator(device="cuda").manual_seed(3)
for label, M, Kd, nnz, hub in (("1e6 x 1e4, row of 1e4", 1_000_000, 10_000, 10_000_000, 10_000),
                               ("2e5 x 1e5, row of 6e4", 200_000, 100_000, 10_000_000, 60_000),
                               ("2e5 x 1e5, row of 1e5", 200_000, 100_000, 10_000_000, 100_000)):
    base = torch.randint(0, M * Kd, (nnz,), device="cuda", generator=g)
    lin = torch.unique(torch.cat([base, torch.randperm(Kd, device="cuda", generator=g)[:hub] + 77 * Kd]))
    vals = torch.rand(lin.numel(), device="cuda") + 0.1
    c = sp.COO._from_sorted_keys(lin, vals, (M, Kd), 0.0, torch.int64)
    row = [label]
    for ca in ((0,), (1,)):
        for n in (16, 128):
            a = sp.GCXS(c, compressed_axes=ca)
            b = torch.rand(Kd, n, device="cuda")
            a @ b
            a @ b
            a @ b
            row.append(f"{'csr' if ca == (0,) else 'csc'} N={n}: {dev_time(lambda: a @ b, 5):.2f}"
                       f"{' split' if a.__dict__.get('_hot_split') is not None else ''}")
    print(" | ".join(row), flush=True)
