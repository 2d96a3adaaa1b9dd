import sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import numpy as np, torch, sparse_amd as sp
from sparse_amd import _ffi
rng = np.random.default_rng(0)
m, n, p, q = 50, 20, 50, 50
t = torch.from_numpy(rng.random((m, n))).cuda()
cases = {"dense.coo": (t, sp.random((m, p, n, q), density=0.01, random_state=1), 1, 2),
         "coo.coo": (sp.random((m, p), density=0.01, random_state=2), sp.random((m, n, p, q), density=0.01, random_state=3), 1, 2),
         "coo.dense": (sp.random((m, n, p, q), density=0.01, random_state=4), t, 1, 1)}
for name, (l, r, li, ri) in cases.items():
    for rt in (np.ndarray, sp.COO):
        f = lambda: sp.tensordot(l, r, axes=([0, li], [0, ri]), return_type=rt)
        for _ in range(10): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 200 * 1e6
        c0 = _ffi.CALLS; f(); calls = _ffi.CALLS - c0
        print(name, rt.__name__, f"{us:.0f} us, {calls} C-ABI calls", flush=True)
name, (l, r, li, ri) = "dense.coo", cases["dense.coo"]
f = lambda: sp.tensordot(l, r, axes=([0, li], [0, ri]), return_type=sp.COO)
names = []
real = _ffi.call
def spy(nm, *a):
    names.append(nm); return real(nm, *a)
_ffi.call = spy
import sparse_amd._kernels as K, sparse_amd._umath as U, sparse_amd._reduce as R
for mod in (K, U, R):
    pass
f(); print(names)
_ffi.call = real
pr = cProfile.Profile(); pr.enable()
for _ in range(200): f()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print("\n".join(s.getvalue().splitlines()[:60]))
