"""fp32 / int32 results of 5..8 columns and fp64 of 3..5: row-group kernel against the executor (padded B), config-2 matrix."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
import sparse_amd as sp
from sparse_amd import _kernels as K, _dot

def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

M, Kd = 1_000_000, 10000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
for dt, ns in ((torch.float32, (5, 6, 7, 8)), (torch.float64, (3, 4, 5))):
    d = data.to(dt)
    a = sp.GCXS((d, idx, ptr), shape=(M, Kd), compressed_axes=(0,))
    _dot.prepare_spmm(a, dt)
    for N in ns:
        b = torch.rand((Kd, N), device="cuda", dtype=dt)
        trg = t(lambda: K.dot_csr_ndarray((M, N), d, idx, ptr, b))
        panel = 64 if dt == torch.float64 else 128
        def ex():
            bp = torch.zeros((Kd, panel), dtype=dt, device="cuda"); bp[:, :N] = b
            return _dot._tiled_product(a, dt, (M, N), Kd, bp)
        tex = t(ex)
        same = torch.equal(ex(), K.dot_csr_ndarray((M, N), d, idx, ptr, b))
        print(f"{str(dt):14s} N={N}: row-group {trg:.3f} ms  executor (padded B) {tex:.3f} ms  identical {same}", flush=True)
