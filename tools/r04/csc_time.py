import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M, Kd = 1_000_000, 10_000
for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
    data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=3, dtype=dt)
    cd, ci, cp = K.csx_swap_2d(data, idx.to(it), ptr.to(it), M, Kd)
    print(dt, f"{t(lambda: K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)):.3f} ms", flush=True)
