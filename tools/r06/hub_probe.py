"""Hub rows: product time with and without the two-part form (`_dot.HOT_ROW_SPLIT`), by width; the parts' own times."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time
from sparse_amd import _dot as D

g = torch.Generator(device="cuda").manual_seed(3)
CASES = [("1e6 x 1e4, row of 1e4", 1_000_000, 10_000, 10_000_000, 10_000), ("1e4 x 1e6, row of 1e6", 10_000, 1_000_000, 10_000_000, 1_000_000),
         ("1e5 x 1e5, row of 1e5", 100_000, 100_000, 10_000_000, 100_000), ("1e6 x 1e6, 8 rows of 3e5", 1_000_000, 1_000_000, 10_000_000, -300_000)]
for label, M, Kd, nnz, hub in CASES:
    base = torch.randint(0, M * Kd, (nnz,), device="cuda", generator=g)
    if hub > 0:
        extra = torch.randperm(Kd, device="cuda", generator=g)[:hub] + 77 * Kd
    else:
        extra = torch.cat([torch.randperm(Kd, device="cuda", generator=g)[:-hub] + r * Kd for r in range(100, 900, 100)])
    lin = torch.unique(torch.cat([base, extra]))
    vals = torch.rand(lin.numel(), device="cuda", dtype=torch.float32) + 0.1
    c = sp.COO._from_sorted_keys(lin, vals, (M, Kd), 0.0, torch.int64)
    for dt in (torch.float32, torch.float64):
        row = [label, str(dt)[6:]]
        for n in (1, 4, 8, 16, 64, 128):
            b = torch.rand(Kd, n, device="cuda", dtype=dt)
            ts = []
            for on in (False, True):
                D.HOT_ROW_SPLIT = on
                a = sp.GCXS(c.astype('float32' if dt == torch.float32 else 'float64'), compressed_axes=(0,))
                a @ b
                a @ b
                ts.append(dev_time(lambda: a @ b, 3))
            sp_ = a.__dict__.get("_hot_split")
            parts = ""
            if sp_ is not None and n in (16, 128):
                tl = dev_time(lambda: D._gcxs_times_dense(sp_[0], b, (M, n)), 3)
                th = dev_time(lambda: D._gcxs_times_dense(sp_[1], b, (int(sp_[1].shape[0]), n)), 3)
                parts = f" [light {tl:.2f} hot {th:.2f} V={sp_[1].shape[0]}]"
            row.append(f"N={n}: {ts[0]:.2f} -> {ts[1]:.2f}{parts}")
        print(" | ".join(row), flush=True)
    D.HOT_ROW_SPLIT = True
