"""a @ a.T with a hub row of 5e4 in a 1e5 x 1e5 matrix of 10 per row, five times, for a kernel trace."""
import sys

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time

g = torch.Generator(device="cuda").manual_seed(3)
n, per = 100_000, 10
base = torch.randint(0, n * n, (n * per,), device="cuda", generator=g)
hubrow = torch.randperm(n, device="cuda", generator=g)[:50_000] + 77 * n
lin = torch.unique(torch.cat([base, hubrow]))
vals = torch.rand(lin.numel(), device="cuda") + 0.1
a = sp.GCXS(sp.COO._from_sorted_keys(lin, vals, (n, n), 0.0, torch.int64), compressed_axes=(0,))
which = sys.argv[1] if len(sys.argv) > 1 else "aat"
f = (lambda: a @ a.T) if which == "aat" else (lambda: a @ a)
f()
print(which, dev_time(f, 5))
