"""N-D matmul: block-diagonal single product vs the per-slice loop.  python tools/batched_time.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _settings
from sparse_amd._batched import matmul_batched, matmul_blockdiag
_settings.NAN_CHECK = False
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for B, M, Kd, N, dens in ((64, 4096, 4096, 128, 0.01), (512, 512, 512, 64, 0.02), (8, 65536, 4096, 128, 0.005)):
    a = sp.random((B, M, Kd), density=dens, random_state=1, dtype=np.float32, idx_dtype=np.int32)
    b = torch.rand((B, Kd, N), device="cuda", dtype=torch.float32)
    t1 = timed(lambda: matmul_blockdiag(a, b))
    t0 = timed(lambda: matmul_batched(a, b), reps=2)
    fl = 2.0 * a.nnz * N
    print(f"batch {B} x ({M}x{Kd} @{dens}) x ({Kd}x{N}): loop {t0:.2f} ms, block-diagonal {t1:.2f} ms ({fl / t1 / 1e6:.0f} GFLOP/s), x{t0 / t1:.1f}")
