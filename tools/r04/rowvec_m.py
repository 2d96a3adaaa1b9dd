"""Row-vector kernel: B resident in LDS against gathered, by row count (SPAMD_ROWVEC_LDS_MIN_M); run with the default library and
with a -DSPAMD_ROWVEC_LDS_MIN_M=0 build (SPAMD_LIB)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for Kd, N in ((10000, 1), (10000, 4), (2000, 2)):
    for M in (2048, 4096, 8192, 16384, 32768, 65536):
        data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1)
        b = torch.rand((Kd, N), device="cuda")
        print(f"K={Kd} N={N} M={M:6d}: {t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b)):.4f} ms", flush=True)
