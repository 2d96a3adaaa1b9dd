"""Row-count crossover of the executor against the row-group kernel after round 4's 1-D grid (sets _dot._tiled_eligible's M bound)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K

def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for dt in (torch.float32, torch.float64):
    for N in (128, 512):
        for M in (4096, 8192, 16384, 24576, 32768, 40960, 50000, 65536):
            Kd, dens = 10000, 0.01
            data, idx, ptr = make_csr_device(M, Kd, dens, seed=1)
            data = data.to(dt)
            b = torch.rand((Kd, N), device="cuda", dtype=dt)
            layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
            out = torch.empty((M, N), device="cuda", dtype=dt)
            tt = t(lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out))
            tr = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=out))
            print(f"{str(dt):14s} N={N:4d} M={M:6d}: tiled {tt:7.4f} ms  rowgroup {tr:7.4f} ms  ratio {tr/tt:5.2f}", flush=True)
