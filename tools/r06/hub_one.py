"""One hub-row product (1e4 x 1e6 with a row of 1e6, N = 16) ten times, for a kernel trace; wall time of the API layers."""
import sys
import time

sys.path.insert(0, "/root/repo")
import torch

import sparse_amd as sp
from bench import dev_time
from sparse_amd import _dot as D

g = torch.Generator(device="cuda").manual_seed(3)
M, Kd, n = 10_000, 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 16
base = torch.randint(0, M * Kd, (10_000_000,), device="cuda", generator=g)
lin = torch.unique(torch.cat([base, torch.randperm(Kd, device="cuda", generator=g) + 77 * Kd]))
vals = torch.rand(lin.numel(), device="cuda") + 0.1
a = sp.GCXS(sp.COO._from_sorted_keys(lin, vals, (M, Kd), 0.0, torch.int64), compressed_axes=(0,))
b = torch.rand(Kd, n, device="cuda")
for _ in range(3):
    a @ b
print("a @ b", dev_time(lambda: a @ b, 10))
print("_dot", dev_time(lambda: D._dot(a, b), 10))
print("_gcxs_times_dense", dev_time(lambda: D._gcxs_times_dense(a, b, (M, n)), 10))
s = a.__dict__["_hot_split"]
print("light", dev_time(lambda: D._gcxs_times_dense(s[0], b, (M, n)), 10), "hot", dev_time(lambda: D._gcxs_times_dense(s[1], b, (int(s[1].shape[0]), n)), 10))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    a @ b
torch.cuda.synchronize()
print("wall per call", (time.perf_counter() - t) * 100, "ms")
