"""Elementwise operations between sparse operands of different shapes (broadcasting, _umath.py:392-751 in the reference) and
three-operand `where`: ms per call at 10^7 stored elements."""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from bench import dev_time

M, Kd = 100_000, 10_000
g = torch.Generator(device="cuda").manual_seed(1)
lin = torch.unique(torch.randint(0, M * Kd, (10_000_000,), device="cuda", generator=g))
x = sp.COO._from_sorted_keys(lin, torch.rand(lin.numel(), device="cuda", dtype=torch.float64) + 0.1, (M, Kd), 0.0, torch.int64)
x.coords
col = sp.COO.from_numpy((np.random.default_rng(0).random((M, 1)) + 0.1) * (np.random.default_rng(1).random((M, 1)) < 0.7))
row = sp.COO.from_numpy((np.random.default_rng(0).random((1, Kd)) + 0.1) * (np.random.default_rng(1).random((1, Kd)) < 0.7))
vec = sp.COO.from_numpy((np.random.default_rng(0).random((Kd,)) + 0.1) * (np.random.default_rng(1).random((Kd,)) < 0.7))
x3 = x.reshape((100, 1000, Kd))
ops = {
    "x*x": lambda: x * x, "x*col": lambda: x * col, "x*row": lambda: x * row, "x*vec": lambda: x * vec, "col*x": lambda: col * x,
    "x+col (dense-ish result)": None, "x3*row": lambda: x3 * row[None], "x3*x[None]": lambda: x3 * x[:1000][None],
    "where(x>0.5,x,0)": lambda: sp.where(x > 0.5, x, 0), "where(x>.5,x,col)": lambda: sp.where(x > 0.5, x, col * 0),
    "x>0.5": lambda: x > 0.5, "maximum(x,row)": lambda: np.maximum(x, row), "x*2+x": lambda: x * 2 + x,
    "x.clip": lambda: x.clip(0.2, 0.8) if hasattr(x, "clip") else None, "x.round": lambda: x.round(1),
    "col*row (outer)": lambda: col * row,
}
for name, f in ops.items():
    if f is None:
        continue
    try:
        r = f()
        f()
        print(f"{name:28s} {dev_time(f, 3):8.2f} ms   nnz {getattr(r, 'nnz', None)}", flush=True)
    except Exception as e:
        print(f"{name:28s} {type(e).__name__}: {str(e)[:80]}", flush=True)
