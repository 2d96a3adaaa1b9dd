"""BASELINE config 1 (two COO 1000^3, 10^6 stored elements each, f64/int64): wall time per `x + y`, `x * y`,
`z.sum(axis=2)`, `z.sum(axis=0)` against the device time of their kernels (run under
`rocprofv3 --kernel-trace --stats` for the latter), and the top of a cProfile of each loop."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp

x = sp.random((1000, 1000, 1000), nnz=1_000_000, random_state=0)
y = sp.random((1000, 1000, 1000), nnz=1_000_000, random_state=1)
z = x + y
ops = {"add": lambda: x + y, "multiply": lambda: x * y, "sum_axis2": lambda: z.sum(axis=2), "sum_axis0": lambda: z.sum(axis=0)}
which = [a for a in sys.argv[1:] if not a.startswith("--")] or list(ops)
for name in which:
    f = ops[name]
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call (wall)", flush=True)
if "--profile" in sys.argv:
    for name in which:
        if name.startswith("--"):
            continue
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            ops[name]()
        torch.cuda.synchronize()
        pr.disable()
        print("=====", name)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
