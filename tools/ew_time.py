import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
xb = sp.random((1000, 1000, 1000), nnz=nb, random_state=10)
yb = sp.random((1000, 1000, 1000), nnz=nb, random_state=11)
for name, f in (("add", lambda: xb + yb), ("mul", lambda: xb * yb)):
    for _ in range(2): z = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): z = f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name} nnz={nb}: {e0.elapsed_time(e1)/3:.3f} ms  out={z.nnz}")
