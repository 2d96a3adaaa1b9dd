import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K, _ffi
for n, dens in ((1000 * 1000, 0.001), (4096 * 4096, 0.01)):
    d = torch.rand(n, device="cuda")
    d = torch.where(d < dens, d, torch.zeros_like(d)).reshape(int(n ** 0.5), -1)
    for flag in (False, True):
        K.DENSE_NONFILL = flag
        for _ in range(3): x = sp.COO.from_numpy(d)
        c0 = _ffi.CALLS; sp.COO.from_numpy(d); calls = _ffi.CALLS - c0
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): x = sp.COO.from_numpy(d)
        torch.cuda.synchronize()
        print(f"n={n} fused={flag}: {(time.perf_counter() - t) / 30 * 1e6:.1f} us, {calls} calls, nnz {x.nnz}")
