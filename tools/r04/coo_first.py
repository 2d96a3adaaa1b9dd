"""Where the first product of a config-2-sized COO operand spends its time."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _dot, _kernels as K, _settings
from bench import make_csr_device
_settings.NAN_CHECK = False
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234, idx_dtype=torch.int32, device="cuda")
b = torch.rand((Kd, N), device="cuda", dtype=torch.float32)
rows_c = K.csr_to_keys(ptr, torch.zeros_like(idx), M, 1).to(idx.dtype)
coo = sp.COO(torch.stack([rows_c, idx]), data, shape=(M, Kd), has_duplicates=False, sorted=True)
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def first():
    _dot.drop_derived(coo); coo.__dict__.pop("_spmm_uses", None)
    return coo @ b
print("first product          %.3f ms" % t(first))
print("rows_to_indptr         %.3f ms" % t(lambda: K.rows_to_indptr(coo.coords[0], M)))
p = K.rows_to_indptr(coo.coords[0], M)
print("indptr dtype", p.dtype, "coords dtype", coo.coords.dtype)
print("csr_tiled_layout       %.3f ms" % t(lambda: K.csr_tiled_layout(coo.data, coo.coords[1].contiguous(), p, M, Kd, defer_check=True)))
def trip():
    _dot.drop_derived(coo)
    return _dot._csr_triplet(coo)
print("_csr_triplet           %.3f ms" % t(trip))
def prep():
    _dot.drop_derived(coo)
    return _dot.prepare_spmm(coo, torch.float32)
print("prepare_spmm           %.3f ms" % t(prep))
print("steady product         %.3f ms" % t(lambda: coo @ b, 10))
