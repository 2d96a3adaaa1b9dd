"""CSC-native inspector at config 2's size: ms per layout, f32/int32 and f64/int64"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from bench import make_csr_device, dev_time
from sparse_amd import _kernels as K
M, Kd = 1_000_000, 10_000
d, i, p = make_csr_device(M, Kd, 0.01, 1234)
for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
    cd, ci, cp = K.csx_swap_2d(d.to(dt), i, p, M, Kd) if dt == torch.float32 else K._csr_to_csc_any(d.to(dt), i.to(it), p.to(it), M, Kd) if hasattr(K, "_csr_to_csc_any") else K.csx_swap_2d(d.to(dt), i.to(it), p.to(it), M, Kd)
    ci, cp = ci.to(it), cp.to(it)
    for _ in range(3): lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
    ms = dev_time(lambda: K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt), 10)
    b = torch.rand((Kd, 128 if dt == torch.float32 else 64), device="cuda", dtype=dt)
    ok = torch.equal(K.dot_csr_ndarray_tiled(lay, (M, b.shape[1]), Kd, b), K.dot_csr_ndarray((M, b.shape[1]), d.to(dt), i, p, b))
    print(dt, it, f"csc inspector {ms:.3f} ms  product identical to row-group {ok}", flush=True)
