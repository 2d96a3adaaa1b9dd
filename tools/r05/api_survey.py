"""us per call (result complete on the device) and C-ABI calls of the in-scope API at the reference's benchmark sizes:
a quick look for operations that go through long chains of small device calls."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _ffi
rng = np.random.default_rng(0)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
x2 = sp.random((side, side), density=0.001, random_state=rng, format="coo")
y2 = sp.random((side, side), density=0.001, random_state=rng, format="coo")
g0 = x2.asformat("gcxs", compressed_axes=(0,))
g1 = x2.asformat("gcxs", compressed_axes=(1,))
x3 = sp.random((100, 100, 100), density=0.001, random_state=rng, format="coo")
y3 = sp.random((100, 100, 100), density=0.001, random_state=rng, format="coo")
d = np.asarray(rng.random((side, 8)))
ops = {
    "coo.T": lambda: x2.T, "coo.transpose(2,0,1)": lambda: x3.transpose((2, 0, 1)), "coo.transpose(1,2,0)": lambda: x3.transpose((1, 2, 0)),
    "coo.reshape": lambda: x3.reshape((1000, 1000)), "coo->gcxs0": lambda: x2.asformat("gcxs", compressed_axes=(0,)),
    "coo->gcxs1": lambda: x2.asformat("gcxs", compressed_axes=(1,)), "gcxs0->coo": lambda: g0.tocoo(), "gcxs0->gcxs1": lambda: g0.change_compressed_axes((1,)),
    "gcxs.T": lambda: g0.T, "todense": lambda: x2.todense(), "sum()": lambda: x2.sum(), "sum(0)": lambda: x2.sum(axis=0), "sum(1)": lambda: x2.sum(axis=1),
    "sum3(0,1)": lambda: x3.sum(axis=(0, 1)), "sum3(1)": lambda: x3.sum(axis=1), "max(0)": lambda: x2.max(axis=0), "mean(1)": lambda: x2.mean(axis=1),
    "x*2": lambda: x2 * 2, "x+y": lambda: x2 + y2, "x*y": lambda: x2 * y2, "x>y": lambda: x2 > y2, "abs": lambda: abs(x2), "sin": lambda: np.sin(x2),
    "x**2": lambda: x2 ** 2, "where": lambda: sp.where(x2 > 0.5, x2, y2), "x3+y3": lambda: x3 + y3, "x@d": lambda: x2 @ d, "g0@d": lambda: g0 @ d,
    "x.T@d": lambda: x2.T @ d, "concat": lambda: sp.concatenate([x2, y2], axis=0), "stack": lambda: sp.stack([x2, y2]), "x[..]nnz": lambda: x2.nnz,
    "astype": lambda: x2.astype(np.float32), "gcxs+gcxs": lambda: g0 + g0, "gcxs*2": lambda: g0 * 2, "gcxs.sum(0)": lambda: g0.sum(axis=0),
    "tensordot3": lambda: sp.tensordot(x3, y3, axes=([0, 1], [0, 1])), "dot3": lambda: sp.dot(x3, y3.transpose((0, 2, 1))[0] if False else d[:100, :]),
    "einsum": lambda: sp.einsum("ij,jk->ik", x2, y2), "kron-free nan_to_num": lambda: sp.nan_to_num(x2), "isnan": lambda: np.isnan(x2),
    "nansum(0)": lambda: sp.nansum(x2, axis=0), "round": lambda: x2.round(2), "clip": lambda: x2.clip(0.1, 0.9), "conj": lambda: x2.conj(),
}
rows = []
for name, f in ops.items():
    try:
        f(); f()
        c0 = _ffi.CALLS; f(); calls = _ffi.CALLS - c0
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 30
        for _ in range(n): r = f()
        torch.cuda.synchronize()
        rows.append((round((time.perf_counter() - t) / n * 1e6, 1), calls, name))
    except Exception as ex:
        rows.append((-1.0, -1, f"{name}: {type(ex).__name__}: {str(ex)[:80]}"))
for us, calls, name in sorted(rows, reverse=True):
    print(f"{us:9.1f} us  {calls:3d} calls  {name}")
