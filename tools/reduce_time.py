import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
xb = sp.random((1000, 1000, 1000), nnz=nb, random_state=10)
for ax in (2, 0):
    for _ in range(2): s = xb.sum(axis=ax)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): s = xb.sum(axis=ax)
    e1.record(); torch.cuda.synchronize()
    print(f"sum(axis={ax}) nnz={nb}: {e0.elapsed_time(e1)/3:.3f} ms  groups={s.nnz}")
