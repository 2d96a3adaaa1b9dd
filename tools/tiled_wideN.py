import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M, Kd = 500_000, 10_000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1)
for dt in (torch.float32, torch.float64):
    d = data.to(dt)
    layout = K.csr_tiled_layout(d, idx, ptr, M, Kd)
    for N in (128, 256, 512, 1024):
        b = torch.rand((Kd, N), device="cuda", dtype=dt); out = torch.empty((M, N), device="cuda", dtype=dt)
        tt = t(lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out))
        tr = t(lambda: K.dot_csr_ndarray((M, N), d, idx, ptr, b, out=out))
        print(f"{dt} M={M} N={N}: tiled {tt:.3f} ms, row-group {tr:.3f} ms")
