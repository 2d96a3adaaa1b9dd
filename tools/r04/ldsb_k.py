"""K = 576..639 (newly inside the LDS-resident-B kernel's budget): spmm_ldsb against the executor and the row-group kernel."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K, _ffi
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dt in (torch.float32, torch.float64):
    for Kd in (512, 620):
        M, N = 262144, 512 if dt == torch.float32 else 256
        data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1, dtype=dt)
        b = torch.rand((Kd, N), device="cuda", dtype=dt)
        lay = K.csr_tiled_layout(data, idx, ptr, M, Kd)
        a = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b))
        c = t(lambda: K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b))
        same = torch.equal(K.dot_csr_ndarray((M, N), data, idx, ptr, b), K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b))
        print(f"{str(dt):14s} K={Kd} N={N}: spamd_spmm_csr (LDS-resident B) {a:.4f} ms, executor {c:.4f} ms, identical {same}", flush=True)
