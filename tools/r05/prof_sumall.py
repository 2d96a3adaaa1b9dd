import sys, cProfile, pstats, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
x2 = sp.random((1000, 1000), density=0.001, random_state=1, format="coo")
for _ in range(3): x2.sum()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): x2.sum()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
