"""Where does the tiled SpMM beat the row-group kernel?  (sets _dot._tiled_eligible's thresholds)"""
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

N = 128
for M, Kd, dens in ((32768, 10000, 0.01), (65536, 10000, 0.01), (131072, 10000, 0.01), (262144, 10000, 0.01),
                    (1_000_000, 10000, 0.003), (1_000_000, 10000, 0.001), (1_000_000, 1000, 0.01), (1_000_000, 1000, 0.05),
                    (262144, 512, 0.01), (200_000, 50000, 0.002)):
    data, idx, ptr = make_csr_device(M, Kd, dens, seed=1)
    b = torch.rand((Kd, N), device="cuda")
    layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
    out = torch.empty((M, N), device="cuda")
    tt = t(lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out))
    tr = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=out))
    ti = t(lambda: K.csr_tiled_layout(data, idx, ptr, M, Kd), reps=3)
    per_list = data.numel() * 4096 / (M * Kd)
    print(f"M={M:8d} K={Kd:6d} dens={dens:6.3f} nnz={data.numel():10d} entries/list={per_list:6.1f}: tiled {tt:7.3f} ms  rowgroup {tr:7.3f} ms  inspector {ti:6.3f} ms")
