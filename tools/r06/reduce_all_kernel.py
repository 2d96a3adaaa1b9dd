"""spamd_reduce_all alone (HIP events, 50 launches): us and GB/s by size; SPAMD_RA_PIECES / SPAMD_LIB pick the variant."""
import os
import sys

sys.path.insert(0, "/root/repo")
import torch

from bench import dev_time
from sparse_amd import _reduce as R

out = [os.environ.get("SPAMD_RA_PIECES", "-")]
for n in (1_000_000, 10_000_000, 100_000_000):
    for dt in (torch.float32, torch.float64):
        vals = torch.rand(n, device="cuda", dtype=dt)
        k = lambda: R.reduce_all(vals, "add")
        k()
        ms = min(dev_time(k, 50) for _ in range(3))
        out.append(f"{n:.0e}/{str(dt)[-2:]} {ms * 1e3:.1f}us {n * vals.element_size() / ms / 1e6:.0f}GB/s")
print(" | ".join(out), flush=True)
