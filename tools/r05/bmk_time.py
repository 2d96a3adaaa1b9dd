"""Config-5 share through `a @ b` (bitmap SpGEMM): ms per product; with a -DBMK_PROF build (SPAMD_BMK_PROF=1) the cycles of
thread 0 per row and phase.  argv: [reps] [f64]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n, share = 1_000_000, 8
f64 = "f64" in sys.argv
if "nopack" in sys.argv:
    K.SPGEMM_PACK_B = False
if "split" in sys.argv:
    K.SPGEMM_BITMAP_SPLIT = "first"
gB = sp.random((n, n), density=1e-4, random_state=7, dtype=np.float64 if f64 else np.float32,
               idx_dtype=np.int64 if f64 else np.int32, format="gcxs", compressed_axes=(0,))
rows = n // share
p1 = int(gB.indptr[rows])
gA = sp.GCXS((gB.data[:p1].contiguous(), gB.indices[:p1].contiguous(), gB.indptr[:rows + 1].contiguous()), shape=(rows, n), compressed_axes=(0,))
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
c = gA @ gB
c = gA @ gB
ref = (c.data.double().sum().item(), int(c.indices.sum().item()), c.nnz)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): c = gA @ gB
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / reps * 1e3
st = dict(K.SPGEMM_STATS)
pc = st.pop("phase_cycles", None)
print(f"share{' f64' if f64 else ''}: {ms:.2f} ms per product, checks {ref}, stats {st}")
if pc:
    per = [v / 256 / (rows / 256) for v in pc]
    print("cycles per row (thread 0), phases:", [round(v / 1000, 1) for v in per], "sum", round(sum(per) / 1000, 1), "k")
