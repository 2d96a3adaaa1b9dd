"""Random value arrays through spamd_reduce_all (every axis reduced, csrc/group_reduce.hip) against NumPy: exact for integers,
max / min / fmax / fmin (NaNs planted) and the logical ops, 1e-12 (f64) / 1e-5 (f32) relative to sum |v| for float sums; sizes
around the piece unit (512 lanes x 16 bytes x 4), the 256-piece cap and their multiples; pointers 0-3 elements off a 16-byte
boundary; two calls in a row on one workspace.
    python tools/fuzz_reduce_all.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

from sparse_amd import _reduce as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
OPS = {"add": np.add, "multiply": np.multiply, "maximum": np.maximum, "minimum": np.minimum, "fmax": np.fmax, "fmin": np.fmin,
       "logical_or": np.logical_or, "logical_and": np.logical_and}
cases = fails = 0
t_end = time.time() + budget
while time.time() < t_end:
    dt = [np.float32, np.float64, np.int32, np.int64, np.uint8][int(rng.integers(0, 5))]
    unit = 512 * (16 // np.dtype(dt).itemsize) * 4
    r = rng.random()
    if r < 0.4:
        n = int(rng.integers(1, 5000))
    elif r < 0.8:
        n = max(1, int(rng.choice([1, 2, 3, 255, 256, 257, 300, 511, 512, 513])) * unit + int(rng.integers(-3, 4)))
    else:
        n = int(rng.integers(1, 30_000_000))
    off = int(rng.integers(0, 4))
    if dt == np.uint8:
        ops = ["logical_or", "logical_and"]
        v = (rng.random(n + off) < float(rng.choice([0.0, 1e-6, 0.5, 1 - 1e-6, 1.0]))).astype(np.uint8)
    elif np.dtype(dt).kind == "i":
        ops = ["add", "multiply", "maximum", "minimum"]
        v = rng.integers(-1000, 1000, size=n + off).astype(dt)
    else:
        ops = ["add", "multiply", "maximum", "minimum", "fmax", "fmin"]
        v = (rng.random(n + off) * 2 - 0.7).astype(dt)
        if rng.random() < 0.3:
            v[rng.integers(0, n + off, size=int(rng.integers(1, 4)))] = np.nan
    op = str(rng.choice(ops))
    if op == "multiply":
        v = np.where(rng.random(n + off) < 20.0 / (n + off), v, 1).astype(dt) if dt != np.uint8 else v
    d = torch.from_numpy(v).cuda()[off:]
    h = v[off:]
    for rep in range(2):
        g, val, c, ng = R.reduce_all(d, op)
        got = val.cpu().numpy()[0]
        ok = ng.tolist() == [1, 0] and int(c[0]) == n and int(g[0]) == 0
        with np.errstate(all="ignore"):
            if op in ("logical_or", "logical_and"):
                want = OPS[op].reduce(h.astype(bool))
                ok = ok and bool(got) == bool(want)
            elif np.dtype(dt).kind == "i" or op in ("maximum", "minimum", "fmax", "fmin"):
                want = OPS[op].reduce(h, dtype=dt) if op in ("add", "multiply") else OPS[op].reduce(h)     # (the array's own width: NumPy widens sums)
                ok = ok and (got == want or (np.isnan(got) and np.isnan(want)))
            else:
                want = OPS[op].reduce(h.astype(np.float64))
                scale = np.nansum(np.abs(h.astype(np.float64))) if op == "add" else abs(want)
                tol = (1e-12 if dt == np.float64 else 1e-5) * max(scale, 1e-300)
                ok = ok and (abs(float(got) - float(want)) <= tol or (np.isnan(got) and np.isnan(want)))
        if not ok:
            break
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", np.dtype(dt).name, op, n, off, got, want, ng.tolist(), int(c[0]), flush=True)
        if fails > 5:
            break
print(f"fuzz_reduce_all: {cases} cases, {fails} mismatches (seed {seed})")
sys.exit(1 if fails else 0)
