"""Where sum(axis=2) at config 1 spends its ~85 us: cProfile of 2000 calls + device time of the kernel alone."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparse_amd as sp  # noqa: E402
from sparse_amd import _reduce  # noqa: E402

x = sp.random((1000, 1000, 1000), density=0.001, random_state=1)
fn = lambda: x.sum(axis=2)
for _ in range(20):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    fn()
torch.cuda.synchronize()
print(f"x.sum(axis=2): {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call (wall)")
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    fn()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
