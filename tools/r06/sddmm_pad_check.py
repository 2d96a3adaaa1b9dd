"""sparse_amd.sddmm at config 4's mask for inner dimensions without a row-cached kernel: ms per call with and without the padding"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _kernels as K
dev = torch.device("cuda:0")
M = N = 100_000; nnz = 10_000_000
s = sp.random((M, N), nnz=nnz, random_state=3, dtype=np.float32)
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
for dt, Kd in ((torch.float32, 96), (torch.float32, 100), (torch.bfloat16, 192), (torch.bfloat16, 200), (torch.float32, 160), (torch.float32, 320), (torch.float64, 48), (torch.bfloat16, 96)):
    a = (torch.rand(M, Kd, device=dev) - 0.5).to(dt); bt = (torch.rand(N, Kd, device=dev) - 0.5).to(dt)
    sm = s if dt != torch.float64 else s.astype(np.float64)
    out = []
    for lim in (1 << 60, 200_000):
        K.SDDMM_PAD_MIN_NNZ = lim
        ms, r = timeit(lambda: sp.sddmm(sm, a, bt=bt))
        out.append((ms, r))
    d = (out[0][1].data.double() - out[1][1].data.double()).abs().max().item() if out[0][1].nnz == out[1][1].nnz else float('nan')
    print(f"{str(dt)[6:]} K={Kd}: as it was {out[0][0]:.3f} ms, padded {out[1][0]:.3f} ms, max abs difference {d:.2e}", flush=True)
