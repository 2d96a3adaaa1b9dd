"""3-D COO tensordot with a dense matrix (config 3's family) over sizes: ms per call, ns per stored element x column"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _settings
_settings.NAN_CHECK = False
def t(f, reps=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for dt in (np.float32, np.float64):
    for s_, dens in ((128, 0.01), (256, 0.01), (512, 0.01), (640, 0.01), (768, 0.01), (1024, 0.003), (1536, 0.001), (2048, 0.0005)):
        x = sp.random((s_, s_, s_), density=dens, random_state=1, dtype=dt)
        for n in (8, 64, 512):
            w = torch.rand((s_, n), device="cuda", dtype=torch.float32 if dt == np.float32 else torch.float64)
            ms = t(lambda: sp.tensordot(x, w, axes=1))
            print(f"{np.dtype(dt).name} side={s_:5d} nnz={x.nnz:9d} n={n:4d}: {ms:8.3f} ms  {ms * 1e6 / (x.nnz * n):7.4f} ns per element-column", flush=True)
        del x
        torch.cuda.empty_cache()
