"""Host cost of one `sparse_amd.matmul(a, b)` on the cached-executor path: a problem small enough that the launch rate,
not the GPU, bounds the loop; prints ms per product and the top of a cProfile of the same loop."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd
from sparse_amd import _settings
from bench import make_csr_device

_settings.NAN_WARNING = "deferred"
M, K, N = 70_000, 10_000, 128
data, idx, ptr = make_csr_device(M, K, 0.01, seed=1)
a = sparse_amd.GCXS((data, idx, ptr), shape=(M, K), compressed_axes=(0,))
b = torch.rand((K, N), device="cuda")
for _ in range(20):
    sparse_amd.matmul(a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    sparse_amd.matmul(a, b)
host = (time.perf_counter() - t0) / 2000 * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 2000 * 1e3
print(f"host-side enqueue {host:.4f} ms per product, wall {total:.4f} ms per product")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    sparse_amd.matmul(a, b)
pr.disable()
torch.cuda.synchronize()
sparse_amd.flush_warnings()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
