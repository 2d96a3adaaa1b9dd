"""Config 3 (COO 512^3 @ 1 %, tensordot with dense 512 x 512, axes=1): the row-group kernel alone, f32 and f64."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n = 512; nnz = int(n ** 3 * 0.01)
for dt in (np.float32, np.float64):
    x = sp.random((n, n, n), nnz=nnz, random_state=3, dtype=dt, idx_dtype=np.int32)
    x2 = x.reshape((n * n, n))
    d = torch.rand((n, n), device="cuda", dtype=torch.float32 if dt == np.float32 else torch.float64)
    ptr = K.rows_to_indptr(x2.coords[0], n * n); idx = x2.coords[1].contiguous(); out = torch.empty((n * n, n), device="cuda", dtype=d.dtype)
    ref = torch.empty_like(out)
    K.dot_csr_ndarray((n * n, n), x2.data, idx, ptr, d, out=ref, keep_order=True)      # the row-group kernel
    K.dot_csr_ndarray((n * n, n), x2.data, idx, ptr, d, out=out)
    print("identical to the row-group kernel:", bool(torch.equal(out, ref)))
    g = lambda: K.dot_csr_ndarray((n * n, n), x2.data, idx, ptr, d, out=ref, keep_order=True)
    for _ in range(10): g()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g()
    e1.record(); torch.cuda.synchronize()
    print(f"row-group kernel {e0.elapsed_time(e1) / 20:.4f} ms")
    f = lambda: K.dot_csr_ndarray((n * n, n), x2.data, idx, ptr, d, out=out)
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byt = nnz * (8 + d.element_size()) + n * n * d.element_size() * (1 + n)
    print(f"{np.dtype(dt).name}: kernel {ms:.4f} ms = {byt / ms / 1e9:.2f} TB/s algorithmic ({byt / ms / 1e9 / 8 * 100:.1f} % of 8 TB/s)")
