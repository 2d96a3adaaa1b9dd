"""the dense-accumulator kernel (spgemm_small.hip) as a SECOND choice once the row products are known, against the bucket /
bitmap kernels, for results of up to 16 k columns"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _kernels as K
def t(f, reps=3):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
for dtype, idt in ((np.float32, np.int32), (np.float64, np.int64)):
    for n_row, n in ((1000, 1000), (3000, 3000), (8000, 8000), (15000, 15000), (100_000, 3000), (1_000_000, 2000), (300, 15000)):
        for per_row in (2, 10, 30, 100, 300):
            if n_row * per_row * per_row > 2e9 or per_row * 3 > n or (dtype == np.float64 and n > 8000):
                continue
            a = sp.random((n_row, n), density=per_row / n, random_state=7, dtype=dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
            b = sp.random((n, n), density=per_row / n, random_state=8, dtype=dtype, idx_dtype=idt, format="gcxs", compressed_axes=(0,))
            out = []
            ref = None
            for second in (False, True):
                K.SPGEMM_SMALL_SECOND = second
                ms, c = t(lambda: a @ b)
                if ref is None:
                    ref = c
                same = c.nnz == ref.nnz and torch.equal(c.indices.long(), ref.indices.long()) and torch.equal(c.data, ref.data)
                out.append(f"{'second' if second else 'before'}: {ms:8.3f} ms ({K.SPGEMM_STATS.get('kernel'):7s}{'' if same else ' DIFFERS'})")
            fill = a.nnz * per_row / (n_row * n)
            print(f"{np.dtype(dtype).name} {n_row}x{n} nnz/row={per_row:4d} products/cell={fill:7.3f}: " + "   ".join(out), flush=True)
            del a, b, c, ref
            torch.cuda.empty_cache()
