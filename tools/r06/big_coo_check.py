"""COO elementwise / reduction / conversion with more than 2^31 stored elements in the RESULT (two operands of 1.3 x 10^9
elements on the keys 5 i and 3 j of a 65536 x 65536 x 2 array: union 2.34 x 10^9)"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd._coo import COO
dev = torch.device("cuda:0")
n = 1_300_000_000
shape = (65536, 65536, 2)
kx = torch.arange(n, device=dev, dtype=torch.int64) * 5
ky = torch.arange(n, device=dev, dtype=torch.int64) * 3
dx = torch.full((n,), 2.0, device=dev, dtype=torch.float32)
dy = torch.full((n,), 0.5, device=dev, dtype=torch.float32)
x = COO._from_sorted_keys(kx, dx, shape, np.float32(0), torch.int64)
y = COO._from_sorted_keys(ky, dy, shape, np.float32(0), torch.int64)
common = (3 * (n - 1)) // 15 + 1
t0 = time.perf_counter(); z = x + y; torch.cuda.synchronize()
print(f"x + y: nnz {z.nnz} expected {2 * n - common} {'ok' if z.nnz == 2 * n - common else 'WRONG'}  {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
zk = z.linear_loc()
print("  keys ascending:", bool((zk[1:] > zk[:-1]).all()), " sum of values:", float(z.data.double().sum()), "expected", 2.0 * n + 0.5 * n, flush=True)
t0 = time.perf_counter(); m = x * y; torch.cuda.synchronize()
print(f"x * y: nnz {m.nnz} expected {common} {'ok' if m.nnz == common else 'WRONG'}  {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
del m
t0 = time.perf_counter(); s2 = z.sum(axis=2); torch.cuda.synchronize()
print(f"z.sum(axis=2): nnz {s2.nnz}  total {float(s2.data.double().sum())}  {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
t0 = time.perf_counter(); tot = z.sum(); torch.cuda.synchronize()
print(f"z.sum(): {float(tot)}  {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
del s2
t0 = time.perf_counter(); g = z.reshape((65536, 131072)).asformat("gcxs", compressed_axes=(0,)); torch.cuda.synchronize()
print(f"asformat gcxs: nnz {g.nnz} indptr {g.indptr.dtype} last {int(g.indptr[-1])} {'ok' if int(g.indptr[-1]) == z.nnz else 'WRONG'}  {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
