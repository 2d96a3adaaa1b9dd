"""x.sum() / x.max() over every axis: the key-free kernel (spamd_reduce_all) beside the grouped reduce over the keys."""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from bench import dev_time
from sparse_amd import _reduce as R

for n in (10_000, 1_000_000, 10_000_000, 100_000_000):
    for dt in (np.float32, np.float64):
        vals = torch.rand(n, device="cuda", dtype=torch.float64).to(torch.float32 if dt == np.float32 else torch.float64)
        keys = torch.arange(n, device="cuda", dtype=torch.int64) * 3
        x = sp.COO._from_sorted_keys(keys, vals, (3, n), 0, torch.int64)
        row = [n, np.dtype(dt).name]
        for direct in (False, True):
            R.REDUCE_ALL_DIRECT = direct
            for name in ("sum", "max"):
                f = getattr(x, name)
                for _ in range(3):
                    f()
                row.append(f"{name}{'*' if direct else ''} {dev_time(f, 20) * 1e3:.1f} us")
        k = lambda: R.reduce_all(vals, "add")
        row.append(f"kernel {dev_time(k, 50) * 1e3:.1f} us = {n * vals.element_size() / dev_time(k, 50) / 1e6:.0f} GB/s")
        print(*row, flush=True)
        del x, vals, keys
