import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from bench_paths import timed
for nnz in (1_000_000, 10_000_000, 100_000_000):
    x = sp.random((1000, 1000, 1000), nnz=nnz, random_state=0)
    y = sp.random((1000, 1000, 1000), nnz=nnz, random_state=1)
    for name, f in (("add", lambda: x + y), ("mul", lambda: x * y)):
        ms, z = timed(f, reps=3)
        b = 2 * nnz * 32 + z.nnz * 32
        print(f"nnz={nnz:>10} {name}: {ms:8.3f} ms  {b/ms/1e6:8.1f} GB/s  out_nnz={z.nnz}")
    ms, s = timed(lambda: x.sum(axis=2), reps=3)
    print(f"nnz={nnz:>10} sum(axis=2): {ms:8.3f} ms  {nnz*16/ms/1e6:8.1f} GB/s")
    ms, s = timed(lambda: x.sum(axis=0), reps=3)
    print(f"nnz={nnz:>10} sum(axis=0): {ms:8.3f} ms  {nnz*16/ms/1e6:8.1f} GB/s")
