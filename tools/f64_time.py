import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1, dtype=torch.float64)
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dt in (torch.float64, torch.float32):
    d = data.to(dt); b = torch.rand((Kd, N), device="cuda", dtype=dt); out = torch.empty((M, N), device="cuda", dtype=dt)
    layout = K.csr_tiled_layout(d, idx, ptr, M, Kd)
    print(dt, f"row-group {t(lambda: K.dot_csr_ndarray((M, N), d, idx, ptr, b, out=out)):.3f} ms",
          f"tiled fma {t(lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)):.3f} ms",
          f"tiled exact {t(lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out, exact=True)):.3f} ms",
          f"inspector {t(lambda: K.csr_tiled_layout(d, idx, ptr, M, Kd), 3):.3f} ms")
