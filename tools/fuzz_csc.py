"""Random CSC operands through the CSC-native inspector (csrc/spmm_tiled.hip `tl_csc_*`): the product from its block stream
must be the row-group kernel's on the CSR twin bit for bit, and (small cases) the stream itself must be the same bytes from
two runs.  Shapes around the block (560 rows), tile (160 columns), window (1024 elements) and histogram-step (4096 elements)
sizes; empty and full columns, empty leading / trailing column ranges, columns longer than a step, one-row and one-column
matrices, more row groups than the LDS histogram holds (split + count kernels).
    python tools/fuzz_csc.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

from sparse_amd import _kernels as K

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
EDGE_M = [1, 2, 34, 35, 36, 559, 560, 561, 1119, 1120, 1121, 5000, 70_001]
EDGE_K = [1, 2, 39, 40, 41, 159, 160, 161, 319, 320, 321, 641, 2000]


def make(M, Kd, dens, style):
    """CSR arrays (rows sorted, columns ascending inside a row) on the device"""
    if style == "uniform":
        nnz = int(min(M * Kd, max(0, rng.poisson(M * Kd * dens))))
        lin = np.unique(rng.integers(0, M * Kd, size=nnz, dtype=np.int64))
    elif style == "fullcols":      # a few columns held by every row, the rest sparse
        cols = rng.choice(Kd, size=min(Kd, int(rng.integers(1, 4))), replace=False)
        lin = (np.arange(M, dtype=np.int64)[:, None] * Kd + cols[None, :]).ravel()
        extra = rng.integers(0, M * Kd, size=int(M * Kd * dens), dtype=np.int64)
        lin = np.unique(np.concatenate([lin, extra]))
    elif style == "band":          # only a band of the columns is populated: empty columns in front and behind
        lo = int(rng.integers(0, Kd))
        hi = int(rng.integers(lo, Kd)) + 1
        nnz = int(M * (hi - lo) * min(1.0, dens * 4))
        lin = np.unique(rng.integers(0, M, size=nnz, dtype=np.int64) * Kd + rng.integers(lo, hi, size=nnz, dtype=np.int64))
    elif style == "rows":          # only a few rows are populated (most row blocks of a column are empty)
        rows = rng.choice(M, size=min(M, int(rng.integers(1, 6))), replace=False).astype(np.int64)
        nnz = int(len(rows) * Kd * min(1.0, dens * 50)) + 1
        lin = np.unique(rows[rng.integers(0, len(rows), size=nnz)] * Kd + rng.integers(0, Kd, size=nnz, dtype=np.int64))
    else:                          # dense
        lin = np.arange(M * Kd, dtype=np.int64)
    rows, cols = lin // Kd, lin % Kd
    ptr = np.zeros(M + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=M), out=ptr[1:])
    return cols, ptr, len(lin)


cases = fails = 0
t_end = time.time() + budget
while time.time() < t_end:
    r = rng.random()
    if r < 0.5:
        M, Kd = int(rng.choice(EDGE_M)), int(rng.choice(EDGE_K))
    elif r < 0.9:
        M, Kd = int(rng.integers(1, 20_000)), int(rng.integers(1, 3000))
    else:
        M, Kd = int(rng.integers(1_331_000, 1_500_000)), int(rng.integers(1, 200))      # past the LDS histogram
    style = str(rng.choice(["uniform", "uniform", "fullcols", "band", "rows", "dense"]))
    dens = float(10 ** rng.uniform(-4, -0.3))
    if style == "dense" and M * Kd > 2_000_000:
        style = "uniform"
    if M * Kd * dens > 3e7:
        dens = 3e7 / (M * Kd)
    cols, ptr, nnz = make(M, Kd, dens, style)
    if nnz == 0:
        continue
    dt = [torch.float32, torch.float64, torch.int32][int(rng.integers(0, 3))]
    it = [torch.int32, torch.int64][int(rng.integers(0, 2))]
    if dt == torch.int32:
        data = torch.from_numpy(rng.integers(-1000, 1000, size=nnz).astype(np.int32)).to(dev)
    else:
        data = torch.from_numpy((rng.random(nnz) - 0.5).astype(np.float32 if dt == torch.float32 else np.float64)).to(dev)
    idx, p = torch.from_numpy(cols).to(dev).to(it), torch.from_numpy(ptr).to(dev).to(it)
    cd, ci, cp = K.csx_swap_2d(data, idx, p, M, Kd)
    ci, cp = ci.to(it), cp.to(it)
    n = 128 if dt != torch.float64 else 64
    b = torch.randint(-50, 50, (Kd, n), device=dev, dtype=torch.int32) if dt == torch.int32 else \
        torch.rand((Kd, n), device=dev, dtype=dt) - 0.5
    lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
    if lay is None:
        continue
    got = K.dot_csr_ndarray_tiled(lay, (M, n), Kd, b)
    want = K.dot_csr_ndarray((M, n), data, idx, p, b)
    ok = torch.equal(got, want) and int(lay.pending.item()) == 0 if getattr(lay, "pending", None) is not None else torch.equal(got, want)
    if ok and nnz < 200_000:       # the stream is the same bytes from a second run
        lay2 = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
        end = int(lay[1][-1])
        ok = torch.equal(lay[1], lay2[1]) and torch.equal(lay[0][: end * 16], lay2[0][: end * 16])
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", M, Kd, style, dens, dt, it, nnz, flush=True)
        if fails > 5:
            break
print(f"fuzz_csc: {cases} cases, {fails} mismatches (seed {seed})")
sys.exit(1 if fails else 0)
