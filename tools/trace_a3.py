import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _ffi
n = 512
for dt in (np.float32, np.float64):
    x = sp.random((n, n, n), nnz=int(n ** 3 * 0.01), random_state=3, dtype=dt, idx_dtype=np.int32)
    d = torch.rand((n, n), device="cuda", dtype=torch.float32 if dt == np.float32 else torch.float64)
    for _ in range(3): sp.tensordot(x, d, axes=1)
    names = []
    orig = _ffi.call
    def spy(name, *a):
        names.append(name); return orig(name, *a)
    _ffi.call = spy
    import sparse_amd._kernels as K, sparse_amd._dot as D, sparse_amd._umath as U
    for m in (K, D, U):
        if hasattr(m, "_ffi"): pass
    sp.tensordot(x, d, axes=1)
    _ffi.call = orig
    print(np.dtype(dt).name, names)
    torch.cuda.synchronize()
    for label, f in (("tensordot", lambda: sp.tensordot(x, d, axes=1)),):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); print(label, (time.perf_counter() - t0) / 50 * 1e3, "ms wall")
    # host time only (no sync)
    t0 = time.perf_counter()
    for _ in range(50): r = sp.tensordot(x, d, axes=1)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("host issue time per call", (t1 - t0) / 50 * 1e3, "ms")
