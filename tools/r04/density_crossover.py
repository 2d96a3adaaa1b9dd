"""Density crossover of the executor against the row-group kernel by result width (sets _dot._tiled_eligible's density bound)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K

def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

M = 262144
for Kd in (10000, 100000):
    for dt in (torch.float32, torch.float64):
        for N in (128, 512):
            for dens in (0.0002, 0.0005, 0.001, 0.002, 0.003, 0.005):
                if Kd == 100000 and dens > 0.001: continue
                data, idx, ptr = make_csr_device(M, Kd, dens, seed=1)
                data = data.to(dt)
                b = torch.rand((Kd, N), device="cuda", dtype=dt)
                layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
                out = torch.empty((M, N), device="cuda", dtype=dt)
                tt = t(lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out))
                tr = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=out))
                ti = t(lambda: K.csr_tiled_layout(data, idx, ptr, M, Kd), reps=3)
                print(f"K={Kd:6d} {str(dt):14s} N={N:4d} dens={dens:6.4f} per_list={dens*4096:5.2f}: tiled {tt:7.4f}  rowgroup {tr:7.4f}  ratio {tr/tt:5.2f}  inspector {ti:6.3f}", flush=True)
