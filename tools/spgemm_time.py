import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
g = sp.random((100_000, 100_000), density=1e-3, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
for i in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    c = g @ g
    torch.cuda.synchronize(); print(f"run {i}: {(time.perf_counter()-t)*1e3:.1f} ms  nnz={c.nnz}  mem={torch.cuda.max_memory_allocated()/1e9:.1f} GB")
