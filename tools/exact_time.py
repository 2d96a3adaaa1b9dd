import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
for dt in (torch.float32, torch.float64):
    data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1, dtype=dt)
    b = torch.rand((Kd, N), device="cuda", dtype=dt)
    layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
    out = torch.empty((M, N), device="cuda", dtype=dt)
    for exact in (False, True):
        f = lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out, exact=exact)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(dt, "exact" if exact else "fma", f"{e0.elapsed_time(e1)/20:.3f} ms")
    del data, idx, ptr, b, layout, out
