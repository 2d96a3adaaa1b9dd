"""Steady-state rate of the tiled executor's inner loop: few tiles, very long lists."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
N = 128
for M, Kd, dens in ((500_000, 128, 0.5), (500_000, 256, 0.25), (500_000, 1024, 0.0625), (1_000_000, 10_000, 0.01)):
    data, idx, ptr = make_csr_device(M, Kd, dens, seed=1)
    b = torch.rand((Kd, N), device="cuda")
    layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
    out = torch.empty((M, N), device="cuda")
    f = lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nnz = data.numel(); ent = int(layout[1][-1]) * 8
    print(f"M={M} K={Kd} nnz={nnz} entries(with padding)={ent}: {ms:.3f} ms, "
          f"{ms*1e-3*2.4e9*1024/ent:.1f} cycles/entry/SIMD @2.4GHz, write-only floor {M*512/5e12*1e3:.3f} ms")
