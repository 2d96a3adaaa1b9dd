"""CSC inspector where the LDS histogram does not hold the row groups (more than 38 k groups = 1.33 M rows): split + count
kernels.  ms per layout at 4 M x 2500 @ 1 % (10^8 nnz) beside config 2's shape (hist path)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_csr_device, dev_time
from sparse_amd import _kernels as K
for M, Kd in ((1_000_000, 10_000), (4_000_000, 2_500)):
    d, i, p = make_csr_device(M, Kd, 0.01, 1234)
    for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
        cd, ci, cp = K.csx_swap_2d(d.to(dt), i.to(it), p.to(it), M, Kd)
        ci, cp = ci.to(it), cp.to(it)
        for _ in range(3):
            lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
        print(M, Kd, str(dt)[6:], str(it)[6:], f"{dev_time(lambda: K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt), 10):.3f} ms", flush=True)
        del lay, cd, ci, cp
    del d, i, p
