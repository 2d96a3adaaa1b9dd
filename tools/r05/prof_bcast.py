"""Elementwise with broadcasting at the reference's benchmark sizes ((side, 1, side) op (side, side)): C-ABI calls of one
operation by name, and a cProfile of 300 operations."""
import sys, time, collections, cProfile, pstats, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _ffi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(0)
x = sp.random((side, 1, side), density=0.001, random_state=rng, format="coo")
y = sp.random((side, side), density=0.001, random_state=rng, format="coo")
for name, f in (("add", lambda: x + y), ("mul", lambda: x * y)):
    f(); f()
    names = []
    orig = _ffi.call
    def logged(n, *a):
        names.append(n)
        return orig(n, *a)
    _ffi.call = logged
    for mod in list(sys.modules.values()):
        if getattr(mod, "__name__", "").startswith("sparse_amd") and hasattr(mod, "_ffi") and getattr(mod, "_ffi") is _ffi:
            pass
    f()
    _ffi.call = orig
    print(name, len(names), dict(collections.Counter(names)))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): r = f()
    torch.cuda.synchronize()
    print(f"  {(time.perf_counter() - t) / 200 * 1e6:.1f} us per call, out nnz {r.nnz}")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): x + y
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
