"""Row-local SpGEMM only (no cross-check): ms per product at 10^5 x 10^5 @ 0.1 % (10^9 products), for ablation builds."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
g = sp.random((100_000, 100_000), density=1e-3, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
for _ in range(2):
    try: c = g @ g
    except Exception as e: print("err", type(e).__name__, str(e)[:80]); break
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    try: c = g @ g
    except Exception as e: break
torch.cuda.synchronize()
print(f"row-local {(time.perf_counter() - t) / 3 * 1e3:.2f} ms")
