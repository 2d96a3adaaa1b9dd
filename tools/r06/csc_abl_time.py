"""ms per CSC layout at config 2's size (f32/int32, f64/int64): one line, for library variants (SPAMD_LIB)"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_csr_device, dev_time
from sparse_amd import _kernels as K
M, Kd = 1_000_000, 10_000
d, i, p = make_csr_device(M, Kd, 0.01, 1234)
out = []
for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
    cd, ci, cp = K.csx_swap_2d(d.to(dt), i.to(it), p.to(it), M, Kd)
    ci, cp = ci.to(it), cp.to(it)
    for _ in range(3):
        lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
    out.append(f"{str(dt)[6:]}/{str(it)[6:]} {dev_time(lambda: K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt), 20):.3f} ms")
    del lay, cd, ci, cp
print("  ".join(out))
