"""sum(axis=0) of BASELINE config 1 (COO (1000, 1000, 1000), 1e6 stored elements): ms per call, merge vs sort."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
x = sp.random((1000, 1000, 1000), density=0.001, random_state=3, format="coo")
def run(n=50):
    for _ in range(5): r = x.sum(axis=0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = x.sum(axis=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r
for flag in (True, False, True):
    K.LEAD_LAST = flag
    ms, r = run()
    print(f"merge={flag}: {ms:.4f} ms, nnz {r.nnz}, stats {K.LEAD_LAST_STATS}")
K.LEAD_LAST = True
