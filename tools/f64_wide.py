import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd = 500_000, 10_000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1, dtype=torch.float64)
layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
for N in (128, 512):
    b = torch.rand((Kd, N), device="cuda", dtype=torch.float64)
    out = torch.empty((M, N), device="cuda", dtype=torch.float64)
    f = lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"f64 N={N}: {e0.elapsed_time(e1)/10:.3f} ms")
