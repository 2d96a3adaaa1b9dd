import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _settings, _kernels as K
_settings.NAN_CHECK = False
s_, n = 1024, 512
x = sp.random((s_, s_, s_), density=0.003, random_state=1, dtype=np.float32)
w = torch.rand((s_, n), device="cuda", dtype=torch.float32)
for _ in range(3): r = sp.tensordot(x, w, axes=1)
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter(); r = sp.tensordot(x, w, axes=1); torch.cuda.synchronize(); print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
print("result", type(r), getattr(r, "shape", None), getattr(r, "dtype", None))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): r = sp.tensordot(x, w, axes=1); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
