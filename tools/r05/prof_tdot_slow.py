import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _ffi, _kernels as K, _dot
rng = np.random.default_rng(0)
for (m, n, p, q) in ((50, 20, 20, 50), (50, 10, 20, 50)):
    t_dev = torch.from_numpy(rng.random((m, n))).cuda()
    x = sp.random((m, n, p, q), density=0.01, random_state=rng)
    f = lambda: sp.tensordot(x, t_dev, axes=([0, 1], [0, 1]), return_type=sp.COO)
    per = []
    for i in range(12):
        names = []
        orig = _ffi.call
        def logged(nm, *a):
            names.append(nm); return orig(nm, *a)
        _ffi.call = logged
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize()
        per.append((round((time.perf_counter() - t) * 1e6), len(names)))
        _ffi.call = orig
        if per[-1][0] > 5000: print("slow call", i, names)
    print((m, n, p, q), per)
