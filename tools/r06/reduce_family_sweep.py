"""The reductions of the reference's API (N3: _sparse_array.py:372-876, _coo/common.py:334-583) at one size, every axis choice,
COO and GCXS: ms per call (host side included).  Looking for calls that cost a multiple of their neighbours."""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from bench import dev_time, make_csr_device

M, Kd, dens = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (100_000, 10_000, 0.01)
d, i, p = make_csr_device(M, Kd, dens, 7, dtype=torch.float64)
g = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
c = g.tocoo()
print("stored elements", c.nnz, flush=True)
METHODS = ["sum", "mean", "var", "std", "max", "min", "prod", "any", "all"]
FUNCS = ["nansum", "nanmax", "nanmin", "nanmean", "nanprod"]
for label, x in (("coo", c), ("gcxs", g)):
    for ax in (None, 0, 1):
        row = [label, f"axis={ax}"]
        for m in METHODS + FUNCS:
            f = (lambda m=m: getattr(x, m)(axis=ax)) if m in METHODS else (lambda m=m: getattr(sp, m)(x, axis=ax))
            try:
                f()
                f()
                row.append(f"{m} {dev_time(f, 5):.2f}")
            except Exception as e:
                row.append(f"{m} {type(e).__name__}")
        print(*row, flush=True)
