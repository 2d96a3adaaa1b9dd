import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _kernels as K
def t(f, reps=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
for n, per_row in ((1000, 100), (1000, 300), (2000, 50), (2000, 100), (500, 200), (2000, 20), (1000, 40), (2000, 33)):
    for dtype, idt in ((np.float32, np.int32), (np.float64, np.int64)):
        g = sp.random((n, n), density=per_row / n, random_state=7, dtype=dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
        out = []
        for lim in (1 << 16, 1 << 22):
            K.SPGEMM_SMALL_MAX_NNZ = lim
            ms, c = t(lambda: g @ g)
            out.append(f"limit 2^{lim.bit_length() - 1}: {ms:.3f} ms ({K.SPGEMM_STATS.get('kernel')})")
        print(f"{np.dtype(dtype).name} n={n} nnz/row={per_row} A nnz={g.nnz}: " + "   ".join(out), flush=True)
