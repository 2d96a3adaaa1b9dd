"""C-ABI calls of one operation, by name, for a few small-size cases of the reference's benchmarks."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _ffi
rng = np.random.default_rng(0)
m, n, p, q = 10, 10, 20, 10
t_dev = torch.from_numpy(rng.random((m, n))).cuda()
xr = sp.random((m, p, n, q), density=0.01, random_state=rng)
xl = sp.random((m, n, p, q), density=0.01, random_state=rng)
gx = sp.random((100, 1, 100), density=0.001, random_state=rng, format="gcxs")
gy = sp.random((100, 100), density=0.001, random_state=rng, format="gcxs")
cases = {"dense.coo_COO": lambda: sp.tensordot(t_dev, xr, axes=([0, 1], [0, 2]), return_type=sp.COO),
         "coo.dense_COO": lambda: sp.tensordot(xl, t_dev, axes=([0, 1], [0, 1]), return_type=sp.COO),
         "coo.coo_ndarray": lambda: sp.tensordot(sp.random((m, p), density=0.01, random_state=1), xl, axes=([0, 1], [0, 2]), return_type=np.ndarray),
         "gcxs bcast add": lambda: gx + gy}
for name, f in cases.items():
    f(); f(); f()
    names = []
    orig = _ffi.call
    def logged(nm, *a):
        names.append(nm); return orig(nm, *a)
    _ffi.call = logged
    try:
        f()
    finally:
        _ffi.call = orig
    print(name, len(names), names)
