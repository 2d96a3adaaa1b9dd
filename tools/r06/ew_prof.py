import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
shape = (1000, 1000, 1000)
x = sp.random(shape, density=0.01, random_state=1); y = sp.random(shape, density=0.01, random_state=2)
dv = np.random.default_rng(0).random(1000)
for name, f in (("x*dv", lambda: x * dv), ("where", lambda: sp.where(x > 0.5, x, y))):
    for _ in range(3): f()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): f()
    torch.cuda.synchronize()
    pr.disable()
    print("=====", name)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
