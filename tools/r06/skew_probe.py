"""A full column (every row holds column 0) and a full row beside 10^7 uniform elements: the calls whose kernels count, rank or
scatter by column / row, against the same matrix without them.  ms per call."""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from bench import dev_time

M, Kd, NNZ = 1_000_000, 10_000, 10_000_000
g = torch.Generator(device="cuda").manual_seed(3)
base = torch.randint(0, M * Kd, (NNZ,), device="cuda", generator=g)
hotcol = torch.arange(M, device="cuda") * Kd                    # (r, 0) for every r
hotrow = torch.arange(Kd, device="cuda") + 77 * Kd              # (77, c) for every c
for label, lin in (("uniform", torch.unique(base)), ("+full column", torch.unique(torch.cat([base, hotcol]))),
                   ("+full row", torch.unique(torch.cat([base, hotrow]))), ("+both", torch.unique(torch.cat([base, hotcol, hotrow])))):
    vals = torch.rand(lin.numel(), device="cuda", dtype=torch.float32) + 0.1
    c = sp.COO._from_sorted_keys(lin, vals, (M, Kd), 0.0, torch.int64)
    c.coords
    csr = c.asformat("gcxs", compressed_axes=(0,))
    csc = c.asformat("gcxs", compressed_axes=(1,))
    b128 = torch.rand(Kd, 128, device="cuda")
    b1 = torch.rand(Kd, 1, device="cuda")
    bt = torch.rand(M, 16, device="cuda")
    ops = {
        "coo->csr": lambda: sp.GCXS(c, compressed_axes=(0,)),
        "coo->csc": lambda: sp.GCXS(c, compressed_axes=(1,)),
        "csr->csc": lambda: csr.change_compressed_axes((1,)),
        "csc->csr": lambda: csc.change_compressed_axes((0,)),
        "csr@128": lambda: csr @ b128,
        "csr@1": lambda: csr @ b1,
        "csc@128 (fresh)": lambda: sp.GCXS((csc.data, csc.indices, csc.indptr), shape=(M, Kd), compressed_axes=(1,)) @ b128,
        "csr.T@16": lambda: csr.T @ bt,
        "sum0": lambda: c.sum(axis=0),
        "sum1": lambda: c.sum(axis=1),
        "csr.sum0": lambda: csr.sum(axis=0),
        "c*c": lambda: c * c,
        "c+c.T-ish": lambda: c + c,
        "c.T": lambda: c.T.linear_loc(),
        "max0": lambda: c.max(axis=0),
    }
    row = [label]
    for name, f in ops.items():
        try:
            f()
            f()
            row.append(f"{name} {dev_time(f, 3):.2f}")
        except Exception as e:
            row.append(f"{name} {type(e).__name__}:{str(e)[:30]}")
    print(" | ".join(row), flush=True)
    del c, csr, csc
