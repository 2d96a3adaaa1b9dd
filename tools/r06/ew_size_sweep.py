"""elementwise add / multiply, sum over an axis, COO -> GCXS and transposition over sizes 10^4 .. 10^8 stored elements:
ns per stored element, to spot cliffs where the algorithm switches"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
def t(f, reps=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for n in (1e4, 1e5, 1e6, 1e7, 1e8):
    shape = (1000, 1000, 1000) if n <= 1e7 else (2000, 2000, 2000)
    dens = n / np.prod(shape)
    x = sp.random(shape, density=dens, random_state=1); y = sp.random(shape, density=dens, random_state=2)
    row = [f"n={x.nnz:10d}"]
    for name, f in (("add", lambda: x + y), ("mul", lambda: x * y), ("sum2", lambda: x.sum(axis=2)), ("sum0", lambda: x.sum(axis=0)),
                    ("T", lambda: x.transpose((2, 0, 1))), ("gcxs", lambda: sp.GCXS.from_coo(x)), ("reshape", lambda: x.reshape((shape[0] * shape[1], shape[2])))):
        try:
            ms = t(f)
            row.append(f"{name} {ms:8.3f} ms ({ms * 1e6 / x.nnz:7.2f} ns/el)")
        except Exception as e:      # noqa: BLE001
            row.append(f"{name} {type(e).__name__}")
    print("  ".join(row), flush=True)
    del x, y
    torch.cuda.empty_cache()
