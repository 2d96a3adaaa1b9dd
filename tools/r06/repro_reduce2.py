import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
d = np.load("/root/repo/tests/golden/_tmp_fuzz_case.npy")
print(d.shape, d.dtype, np.count_nonzero(d), flush=True)
x = sp.COO.from_numpy(d)
for rep in range(3):
    for name, ax in (("prod", 0), ("sum", 0), ("prod", 1), ("prod", (0, 1))):
        print(name, ax, end=" ", flush=True)
        got = getattr(x, name)(axis=ax)
        torch.cuda.synchronize()
        print("ok" if np.array_equal(got.todense(), getattr(d, name)(axis=ax)) else "MISMATCH", "fill", got.fill_value, "nnz", got.nnz, flush=True)
