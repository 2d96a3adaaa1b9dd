"""a spread of elementwise calls at 10^7 stored elements of a 1000^3 COO: ms per call and host-fallback counters"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
def t(f, reps=4):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
shape = (1000, 1000, 1000)
x = sp.random(shape, density=0.01, random_state=1); y = sp.random(shape, density=0.01, random_state=2)
v = sp.random((1000,), density=0.5, random_state=3); m = sp.random((1000, 1000), density=0.1, random_state=4)
g = sp.GCXS.from_coo(x.reshape((1000, 1000_000))); h = sp.GCXS.from_coo(y.reshape((1000, 1000_000)))
dv = np.random.default_rng(0).random(1000)
cases = [("-x", lambda: -x), ("x*2.0", lambda: x * 2.0), ("x+1.0", lambda: x + 1.0), ("abs", lambda: abs(x)), ("sqrt", lambda: np.sqrt(x)),
         ("x>0.5", lambda: x > 0.5), ("x**2", lambda: x ** 2), ("exp", lambda: np.exp(x)), ("x+y", lambda: x + y), ("x*y", lambda: x * y), ("x-y", lambda: x - y),
         ("max(x,y)", lambda: np.maximum(x, y)), ("x==y", lambda: x == y), ("x*v (bcast 1-D sparse)", lambda: x * v), ("x*m (bcast 2-D sparse)", lambda: x * m),
         ("x*dv (dense vector)", lambda: x * dv), ("where(x>0.5,x,y)", lambda: sp.where(x > 0.5, x, y)), ("elemwise lambda a*b+a", lambda: sp.elemwise(lambda a, b: a * b + a, x, y)),
         ("astype f32", lambda: x.astype(np.float32)), ("isnan", lambda: np.isnan(x)), ("g+h (gcxs)", lambda: g + h), ("g*h (gcxs)", lambda: g * h), ("g*2 (gcxs)", lambda: g * 2.0),
         ("x.sum()", lambda: x.sum()), ("x.max(axis=1)", lambda: x.max(axis=1)), ("x.mean(axis=(0,1))", lambda: x.mean(axis=(0, 1))), ("x.any(axis=2)", lambda: x.any(axis=2)),
         ("nansum axis0", lambda: sp.nansum(x, axis=0)), ("x.todense()", lambda: x.todense() if False else None), ("x[5]", lambda: x[5]), ("x[:, 3:700:2]", lambda: x[:, 3:700:2]),
         ("concat", lambda: sp.concatenate([x, y], axis=0)), ("stack", lambda: sp.stack([x, y], axis=0)), ("x.T", lambda: x.T), ("tril-like nonzero", lambda: x.nonzero()),
         ("x.to_scipy? flatten", lambda: x.reshape((-1,))), ("dot x.reshape @ dense", lambda: g @ torch.rand((1000_000, 8), device="cuda", dtype=torch.float64))]
sp.fallback_stats(reset=True)
for name, f in cases:
    try:
        before = dict(sp.fallback_stats())
        ms = t(f)
        after = sp.fallback_stats()
        fb = {k: after[k] - before.get(k, 0) for k in after if after[k] != before.get(k, 0)}
        print(f"{name:32s} {ms:9.3f} ms {('  HOST FALLBACK ' + str(fb)) if fb else ''}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:32s} {type(e).__name__}: {str(e)[:90]}", flush=True)
