"""CSR x dense at very narrow results (N = 1 is SpMV) on the config-2 matrix: row-vector kernel vs the k-ascending
row-group kernel, with a cross-check of the two."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd = 1_000_000, 10_000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=7)
def t(f, reps=10):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
import os
DTS = (torch.float32,) if os.environ.get("NARROW_QUICK") else (torch.float32, torch.float64, torch.int64)
for dt in DTS:
    d = data.to(dt) if dt.is_floating_point else (data * 7).to(dt)
    for N in (1, 2, 3, 4):
        b = torch.rand((Kd, N), device="cuda", dtype=torch.float64)
        b = b.to(dt) if dt.is_floating_point else (b * 9).to(dt)
        out = torch.empty((M, N), device="cuda", dtype=dt); ref = torch.empty_like(out)
        ms = t(lambda: K.dot_csr_ndarray((M, N), d, idx, ptr, b, out=out))
        ms0 = t(lambda: K.dot_csr_ndarray((M, N), d, idx, ptr, b, out=ref, keep_order=True))
        if dt.is_floating_point:
            err = float(((out - ref).abs() / ref.abs().clamp_min(1e-30)).max())
        else:
            err = int((out != ref).sum())
        byt = d.numel() * (d.element_size() + 4) + M * 4 + Kd * N * d.element_size() + M * N * d.element_size()
        print(f"{str(dt)[6:]} N={N}: row-vector {ms:.3f} ms ({byt / ms / 1e9 / 8 * 100:.1f} % of HBM peak), row-group {ms0:.3f} ms, "
              f"max rel diff {err:.2e}", flush=True)
