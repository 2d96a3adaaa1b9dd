import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _umath as U
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(40):
    shape = (int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 300)))
    size = shape[0] * shape[1] * shape[2]
    n1 = int(min(size, rng.integers(0, 3_000_000))); n2 = int(min(size, rng.integers(0, 3_000_000)))
    x = sp.random(shape, nnz=n1, random_state=int(rng.integers(1 << 30)))
    y = sp.random(shape, nnz=n2, random_state=int(rng.integers(1 << 30)))
    for name, f in (("add", lambda a, c: a + c), ("mul", lambda a, c: a * c), ("max", lambda a, c: np.maximum(a, c))):
        U.MERGE_SINGLE_PASS = True
        r1 = f(x, y); torch.cuda.synchronize()
        U.MERGE_SINGLE_PASS = False
        r2 = f(x, y); torch.cuda.synchronize()
        U.MERGE_SINGLE_PASS = True
        k1, k2 = r1.linear_loc(), r2.linear_loc()
        ok = k1.shape == k2.shape and torch.equal(k1, k2) and torch.equal(r1.data, r2.data)
        if not ok:
            bad += 1
            msg = f"MISMATCH {name} shape={shape} size={size} n1={x.nnz} n2={y.nnz} out1={r1.nnz} out2={r2.nnz}"
            if k1.shape == k2.shape:
                dk = (k1 != k2).nonzero().flatten(); dv = (r1.data != r2.data).nonzero().flatten()
                msg += f" keydiff={dk.numel()} first={dk[:3].tolist()} valdiff={dv.numel()} first={dv[:3].tolist()}"
                if dv.numel():
                    i = int(dv[0]); msg += f" v1={r1.data[i].item()} v2={r2.data[i].item()}"
            print(msg)
print("bad", bad)
