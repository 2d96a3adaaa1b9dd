import sys, itertools
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
rng = np.random.default_rng(5)
shape = (141, 64, 64, 2)
d = rng.integers(-9, 10, shape).astype(np.int64)
x = sp.COO.from_numpy(d)
print("nnz", x.nnz, flush=True)
for name in ("sum", "max", "min", "prod"):
    for ax in [None] + list(range(4)) + list(itertools.combinations(range(4), 2)):
        print(name, ax, end=" ", flush=True)
        got = getattr(x, name)(axis=ax)
        torch.cuda.synchronize()
        want = getattr(d, name)(axis=ax)
        g = got.todense() if hasattr(got, "todense") else np.asarray(got)
        print("ok" if np.array_equal(g, want) else "MISMATCH", flush=True)
