"""Reductions at config 1's size (10^6 stored elements of a 1000^3 COO): ms per call, C-ABI calls per call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparse_amd as sp  # noqa: E402
from bench_paths import overheads, timed  # noqa: E402
from sparse_amd import _reduce  # noqa: E402

x = sp.random((1000, 1000, 1000), density=0.001, random_state=1)
for ax in (2, 0, (0, 1), None):
    for label, kw in (("", {}),):
        fn = lambda: _reduce.reduce_impl(x, np.add, axis=ax, **kw)
        ms, _ = timed(fn, reps=50, warm=5)
        print(f"sum(axis={ax}) {label:13s}: {ms * 1e3:7.1f} us   {overheads(fn)}", flush=True)
