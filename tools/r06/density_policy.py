"""executor vs row-group kernel around the density threshold of `_dot._tiled_min_density` (N = 128 fp32 / fp64), larger K than
round 4's measurements (K = 10^4 and 10^5 at M = 262144)"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_csr_device, dev_time
from sparse_amd import _kernels as K, _dot
N = 128
for dt in (torch.float32, torch.float64):
    for M, Kd, nnz in ((1_000_000, 20_000, 3e7), (1_000_000, 40_000, 3e7), (1_000_000, 20_000, 6e7), (1_000_000, 40_000, 6e7), (1_000_000, 10_000, 1.5e7),
                       (1_000_000, 10_000, 2e7), (300_000, 10_000, 3e6), (300_000, 10_000, 6e6), (300_000, 20_000, 6e6), (300_000, 20_000, 1.2e7),
                       (300_000, 40_000, 1.2e7), (1_000_000, 5_000, 1e7), (1_000_000, 2_000, 4e6), (1_000_000, 2_000, 8e6)):
        d, i, p = make_csr_device(M, Kd, nnz / (M * Kd), 5, dtype=dt)
        b = torch.rand((Kd, N), device="cuda", dtype=dt)
        ms_g = dev_time(lambda: K.dot_csr_ndarray((M, N), d, i, p, b), 5)
        lay = K.csr_tiled_layout(d, i, p, M, Kd, dtype=dt)
        n = N if dt == torch.float32 else N
        ms_t = dev_time(lambda: K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b), 5)
        per = d.numel() * 4096 / (M * Kd)
        need = _dot._tiled_min_density(N * d.element_size(), Kd * N * d.element_size())
        print(f"{str(dt)[6:]} M={M} K={Kd} nnz={d.numel()}: per-4096 {per:5.2f} (policy needs {need}: {'tiled' if per >= need else 'general'})  "
              f"executor {ms_t:.3f} ms  row-group {ms_g:.3f} ms  {'POLICY WRONG' if (per >= need) != (ms_t < ms_g) else ''}", flush=True)
        del d, i, p, b, lay
        torch.cuda.empty_cache()
