"""Random A @ B (both compressed by rows) through the product path against the same product with the dense-accumulator kernel
switched off (bucket / bitmap / global kernels): indptr, indices and values bit for bit.  Shapes on both sides of the
accumulator kernel's limits (columns, one product per cell, A's size), rectangular operands, empty rows, all four value types.
    python tools/fuzz_spgemm.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import sparse_amd as sp
from sparse_amd import _kernels as K

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
cases = fails = 0
took = {}
t_end = time.time() + budget
while time.time() < t_end:
    n_row = int(rng.choice([1, 7, 300, 1000, 4000, 20_000, 100_000]))
    n_in = int(rng.choice([1, 50, 900, 3000, 9000]))
    n_col = int(rng.choice([1, 40, 1000, 2000, 2100, 3900, 4000, 8000, 8100, 15_000, 17_000, 40_000]))
    pa = float(rng.choice([0.5, 3, 20, 80])) 
    pb = float(rng.choice([0.5, 3, 20, 80, 300]))
    da, db = min(1.0, pa / n_in), min(1.0, pb / n_col)
    if n_row * n_in * da * n_col * db > 6e7 or n_row * n_in * da > 4e6 or n_in * n_col * db > 4e6:
        continue
    dtype = [np.float32, np.float64, np.int32, np.int64][int(rng.integers(0, 4))]
    idt = np.int32 if rng.random() < 0.5 else np.int64
    kw = dict(dtype=np.float64 if np.dtype(dtype).kind == "i" else dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
    a = sp.random((n_row, n_in), density=da, random_state=int(rng.integers(1 << 30)), **kw)
    b = sp.random((n_in, n_col), density=db, random_state=int(rng.integers(1 << 30)), **kw)
    if np.dtype(dtype).kind == "i":
        tdt = torch.int32 if dtype == np.int32 else torch.int64
        a = sp.GCXS(((a.data * 200 - 100).to(tdt), a.indices, a.indptr), shape=a.shape, compressed_axes=(0,))
        b = sp.GCXS(((b.data * 200 - 100).to(tdt), b.indices, b.indptr), shape=b.shape, compressed_axes=(0,))
    c = a @ b
    kern = K.SPGEMM_STATS.get("kernel") if a.nnz and b.nnz else "empty"
    took[kern] = took.get(kern, 0) + 1
    K.SPGEMM_SMALL = False
    try:
        ref = a @ b
    finally:
        K.SPGEMM_SMALL = True
    ok = (c.shape == ref.shape and c.nnz == ref.nnz and torch.equal(c.indptr.long(), ref.indptr.long())
          and torch.equal(c.indices.long(), ref.indices.long()) and torch.equal(c.data, ref.data))
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", n_row, n_in, n_col, pa, pb, dtype, idt, kern, flush=True)
        if fails > 5:
            break
print(f"fuzz_spgemm: {cases} cases, {fails} mismatches (seed {seed}); kernels taken: {took}")
sys.exit(1 if fails else 0)
