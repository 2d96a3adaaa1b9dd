"""Steady-state decomposition of the tiled executor (float32): long lists (K = 128, one tile, no barriers in the
loop) and the headline shape, under the timing ablations of SPAMD_TILED_DBG (set by the caller):
   0 full, 5 no fma (P2 skipped), 6 no LDS reads and no fma (scalar stream + loop only), 2 no tile DMA."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
N = 128
for M, Kd, dens in ((1_000_000, 10_000, 0.01),):
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
    ent = int(layout[1][-1]) * 8
    print(f"DBG={os.environ.get('SPAMD_TILED_DBG', '0')} M={M} K={Kd}: {ms:.3f} ms, {ms*1e-3*2.4e9*1024/ent:.1f} cycles/entry/SIMD")
