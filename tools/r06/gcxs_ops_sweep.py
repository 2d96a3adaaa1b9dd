"""GCXS (2-D, 3 x 10^7 stored elements) through the API: elementwise, reductions, conversions, slicing - ms per call"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
def t(f, reps=4):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
M, Kd = 100_000, 30_000
g = sp.random((M, Kd), density=0.01, random_state=1, format="gcxs", compressed_axes=(0,))
h = sp.random((M, Kd), density=0.01, random_state=2, format="gcxs", compressed_axes=(0,))
hc = h.change_compressed_axes((1,))
c = g.asformat("coo")
v = np.random.default_rng(0).random(Kd) + 0.5
cases = [("g+h same axes", lambda: g + h), ("g*h same axes", lambda: g * h), ("g+hc (csr + csc)", lambda: g + hc), ("g*hc", lambda: g * hc), ("g+coo", lambda: g + c),
         ("g*2", lambda: g * 2.0), ("abs(g)", lambda: abs(g)), ("g*v (dense row vector)", lambda: g * v), ("g*v[:M,None]", lambda: g * (np.arange(M) + 1.0)[:, None]),
         ("g.sum()", lambda: g.sum()), ("g.sum(axis=0)", lambda: g.sum(axis=0)), ("g.sum(axis=1)", lambda: g.sum(axis=1)), ("hc.sum(axis=0)", lambda: hc.sum(axis=0)), ("hc.sum(axis=1)", lambda: hc.sum(axis=1)),
         ("g.max(axis=1)", lambda: g.max(axis=1)), ("g.mean(axis=0)", lambda: g.mean(axis=0)), ("g.T", lambda: g.T), ("g.T.T", lambda: g.T.T),
         ("g.change_compressed_axes((1,))", lambda: g.change_compressed_axes((1,))), ("g.asformat(coo)", lambda: g.asformat("coo")), ("coo.asformat(gcxs)", lambda: c.asformat("gcxs")),
         ("g.tocoo fresh obj", lambda: sp.GCXS((g.data, g.indices, g.indptr), shape=g.shape, compressed_axes=(0,)).asformat("coo")),
         ("g[500:60000]", lambda: g[500:60000]), ("g[:, 100:20000]", lambda: g[:, 100:20000]), ("g[::3]", lambda: g[::3]), ("g[123]", lambda: g[123]), ("g[:, 77]", lambda: g[:, 77]),
         ("g.reshape((M*3, Kd//3))", lambda: g.reshape((M * 3, Kd // 3))), ("g.astype(f32)", lambda: g.astype(np.float32)), ("g == h", lambda: g == h),
         ("g.nonzero()", lambda: g.nonzero()), ("g.copy()", lambda: g.copy()), ("stack/concat [g,h] axis0", lambda: sp.concatenate([g, h], axis=0)), ("concat axis1", lambda: sp.concatenate([g, h], axis=1)),
         ("g @ g.T? skip", None), ("g.todense small? skip", None), ("np.isnan(g).any()", lambda: np.isnan(g).any()), ("g.nnz", lambda: g.nnz), ("g.density", lambda: g.density)]
for name, f in cases:
    if f is None: continue
    try:
        print(f"{name:36s} {t(f):9.3f} ms", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:36s} {type(e).__name__}: {str(e)[:100]}", flush=True)
