"""CSR x dense with a short contracted axis: the LDS-resident-B kernel (spmm_ldsb.hip) against the row-group kernel over
row lengths, result widths and dtypes (the dispatcher's policy: K <= 575, N * itemsize >= 128, M >= 8192)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
def t(f, reps=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, Kd, dens, N, dt in ((262144, 512, 0.01, 512, torch.float32), (262144, 512, 0.01, 128, torch.float32), (262144, 512, 0.01, 32, torch.float32),
                           (262144, 512, 0.05, 512, torch.float32), (262144, 512, 0.2, 128, torch.float32), (262144, 512, 0.2, 512, torch.float32),
                           (262144, 128, 0.04, 256, torch.float32), (1_000_000, 64, 0.1, 64, torch.float32), (20000, 512, 0.02, 512, torch.float32),
                           (262144, 512, 0.01, 512, torch.float64), (262144, 512, 0.2, 128, torch.float64), (262144, 256, 0.02, 64, torch.float64),
                           (262144, 512, 0.01, 512, torch.int32)):
    data, idx, ptr = make_csr_device(M, Kd, dens, seed=5)
    data = (data * 50).to(dt) if not dt.is_floating_point else data.to(dt)
    b = (torch.rand((Kd, N), device="cuda") * 4).to(dt)
    o1 = torch.empty((M, N), device="cuda", dtype=dt); o2 = torch.empty_like(o1)
    a = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=o1))
    r = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=o2, keep_order=True))
    print(f"M={M} K={Kd} nnz/row={Kd * dens:.0f} N={N} {str(dt)[6:]}: LDS-B {a:.3f} ms, row-group {r:.3f} ms ({r / a:.2f}x), equal {bool(torch.equal(o1, o2))}", flush=True)
