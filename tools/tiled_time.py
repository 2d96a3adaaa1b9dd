import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234)
b = torch.rand((Kd, N), device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
torch.cuda.synchronize(); print(f"inspector: {(time.perf_counter()-t)*1e3:.1f} ms")
out = torch.empty((M, N), device="cuda")
for name, f in (("tiled", lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)),
                ("rowgroup", lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=out))):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/10:.3f} ms")
a = K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b)
c = K.dot_csr_ndarray((M, N), data, idx, ptr, b)
print("bit-identical:", torch.equal(a, c))
