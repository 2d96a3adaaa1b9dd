"""Fused single-launch union (every tile searches its own diagonals) against partition kernel + single pass, by total items
(sets _umath.MERGE_FUSED_MAX_ITEMS)."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_amd import _umath as U

dev = torch.device("cuda")
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

z = np.float64(0)
for lg in (20, 21, 22, 23, 24, 25, 26):
    n = 1 << (lg - 1)
    ka = torch.arange(0, n, device=dev) * 3
    kb = ka + (torch.arange(0, n, device=dev) % 2)
    va = torch.randn(n, device=dev, dtype=torch.float64); vb = torch.randn(n, device=dev, dtype=torch.float64)
    out = []
    for fused_max in (1 << 40, 0):
        U.MERGE_FUSED_MAX_ITEMS = fused_max
        out.append(t(lambda: U.merge_union("add", ka, va, kb, vb, z, z, z)))
    print(f"2^{lg} items: fused {out[0]:.4f} ms, partition + single pass {out[1]:.4f} ms", flush=True)
