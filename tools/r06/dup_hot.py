"""COO construction with one hot coordinate (many duplicates of it among unique ones): time of the canonicalisation."""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp

for hot, uniq in ((1000, 1_000_000), (100_000, 1_000_000), (1_000_000, 1_000_000), (10_000_000, 100), (10_000_000, 10_000_000)):
    keys = torch.cat([torch.zeros(hot, dtype=torch.int64, device="cuda"), torch.arange(1, uniq + 1, device="cuda") * 3])
    coords = torch.stack([keys // 100_000, keys % 100_000])
    vals = torch.ones(keys.numel(), device="cuda", dtype=torch.float64)
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        c = sp.COO(coords, vals, shape=(400_000, 100_000))
        s = float(c.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(hot, uniq, f"{dt * 1e3:.2f} ms", c.nnz, s, flush=True)
