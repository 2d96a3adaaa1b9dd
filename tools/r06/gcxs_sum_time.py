"""GCXS x.sum() / x.max() over every axis (per call, incl. the host side)."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time, make_csr_device
from sparse_amd import _reduce as R

for M, Kd, dens in ((10_000, 10_000, 0.01), (1_000_000, 10_000, 0.001), (1_000_000, 10_000, 0.01)):
    d, i, p = make_csr_device(M, Kd, dens, 7, dtype=torch.float64)
    x = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
    row = [M, Kd, int(d.numel())]
    for direct in (False, True):
        R.REDUCE_ALL_DIRECT = direct
        for name in ("sum", "max"):
            f = getattr(x, name)
            for _ in range(3):
                f()
            row.append(f"{name}{'*' if direct else ''} {dev_time(f, 20) * 1e3:.1f} us")
    print(*row, flush=True)
