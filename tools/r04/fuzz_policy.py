"""Randomised `a @ dense` around the executor's policy bounds (rows, density, width, dtype, COO / CSR / CSC operands): whatever
kernel the policy picks, the result must be the row-group kernel's bit for bit (k-ascending FMA per output element) -
except results of at most 4 columns (row-vector kernel: tree order), compared within rounding.
    python tools/r04/fuzz_policy.py [seconds] [seed]"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
import sparse_amd as sp
from sparse_amd import _kernels as K, _dot

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
t_end = time.time() + budget
it = took = 0
while time.time() < t_end:
    dt = [torch.float32, torch.float64, torch.int32][rng.integers(0, 3)]
    M = int(rng.choice([3000, 4100, 5200, 8200, 10300, 16400, 20500, 33000, 41000, 45100, 50000, 66000, 90000]))
    Kd = int(rng.choice([600, 700, 1500, 4000, 10000, 20000]))
    N = int(rng.choice([5, 6, 7, 8, 13, 32, 63, 64, 65, 127, 128, 129, 256, 300, 512, 640]))
    dens = float(rng.choice([0.0006, 0.0011, 0.0013, 0.0016, 0.0023, 0.003, 0.01, 0.02]))
    if M * Kd * dens > 3e7 or M * N > 6e7:
        continue
    data, idx, ptr = make_csr_device(M, Kd, dens, seed=int(rng.integers(0, 1 << 30)), dtype=torch.float64 if dt == torch.float64 else torch.float32)
    if dt == torch.int32:
        data = (data * 2000 - 1000).to(torch.int32)
        b = torch.randint(-3000, 3000, (Kd, N), device="cuda", dtype=torch.int32)
    else:
        data = data - 0.5
        b = torch.rand((Kd, N), device="cuda", dtype=dt) - 0.5
    a = sp.GCXS((data, idx, ptr), shape=(M, Kd), compressed_axes=(0,))
    kind = int(rng.integers(0, 4))
    if kind == 1:
        a = a.change_compressed_axes((1,))
    elif kind == 2:
        a = a.tocoo()
    want = K.dot_csr_ndarray((M, N), data, idx, ptr, b)
    flip = kind != 2 and int(rng.integers(0, 4)) == 0     # dense @ sparse: (b^T a^T)^T through the same kernels
    at, bt_ = (a.T, b.t().contiguous()) if flip else (None, None)
    for rep in range(2):
        got = (bt_ @ at).t() if flip else a @ b
        if not torch.equal(got, want):
            print("MISMATCH", dt, M, Kd, N, dens, kind, rep, bool(getattr(a, "_tiled_layouts", None)), float((got.double() - want.double()).abs().max()))
            sys.exit(1)
    took += bool(getattr(a, "_tiled_layouts", None)) or bool(flip and getattr(at.__dict__.get("_t_view"), "_tiled_layouts", None))
    it += 1
print(f"fuzz_policy ok: {it} products, {took} through the executor")
