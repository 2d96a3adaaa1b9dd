import sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch, sparse_amd as sp
for fmt in ("gcxs", "coo"):
    x = sp.random((1000, 1000), density=0.01, random_state=1, format=fmt)
    y = sp.random((1000, 1000), density=0.01, random_state=2, format=fmt)
    for _ in range(20): x @ y
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): x @ y
    torch.cuda.synchronize()
    print(fmt, "us per call", (time.perf_counter() - t0) / 300 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): x @ y
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32); print("\n".join(s.getvalue().splitlines()[:50]))
