import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _settings
_settings.NAN_CHECK = False
for dt, s_, dens, n in ((np.float64, 1024, 0.003, 512), (np.float32, 2048, 0.0005, 512), (np.float64, 2048, 0.0005, 512), (np.float32, 1536, 0.001, 512)):
    x = sp.random((s_, s_, s_), density=dens, random_state=1, dtype=dt)
    w = torch.rand((s_, n), device="cuda", dtype=torch.float32 if dt == np.float32 else torch.float64)
    ts = []
    for i in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = sp.tensordot(x, w, axes=1); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(np.dtype(dt).name, s_, n, "out GB", round(r.numel() * r.element_size() / 1e9, 2), " ms per call:", " ".join(f"{v:.2f}" for v in ts),
          " reserved GB", round(torch.cuda.memory_reserved() / 1e9, 1), flush=True)
    del x, w, r
    torch.cuda.empty_cache()
