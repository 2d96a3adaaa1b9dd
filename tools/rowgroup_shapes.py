"""The shapes the inspector/executor kernel does not take by policy: row-group kernel vs the tiled
executor forced (B zero-padded to a whole panel where needed), ms per product."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K

def t(f, reps=10):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

import os
SHAPES = ((1_000_000, 10_000, 32, torch.float32), (1_000_000, 10_000, 64, torch.float32), (50_000, 10_000, 128, torch.float32),
          (20_000, 10_000, 128, torch.float32), (1_000_000, 10_000, 128, torch.int32), (1_000_000, 10_000, 16, torch.float64))
if os.environ.get("NARROW"):     # widths between the row-vector kernel (<= 4) and the round-3 executor policy
    SHAPES = tuple((1_000_000, 10_000, n, torch.float32) for n in (5, 8, 16, 24)) + tuple((1_000_000, 10_000, n, torch.float64) for n in (5, 8, 12))
for M, Kd, N, dt in SHAPES:
    data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=7)
    if dt in (torch.int32,):
        data = (data * 100).to(dt)
    else:
        data = data.to(dt)
    b = (torch.rand((Kd, N), device="cuda") * 10).to(dt)
    out = torch.empty((M, N), device="cuda", dtype=dt)
    rg = t(lambda: K.dot_csr_ndarray((M, N), data, idx, ptr, b, out=out))
    line = f"M={M} K={Kd} N={N} {str(dt)[6:]}: row-group {rg:.3f} ms"
    if dt in (torch.float32, torch.float64):
        panel = 128 if dt == torch.float32 else 64
        npad = -(-N // panel) * panel
        bp = torch.zeros((Kd, npad), device="cuda", dtype=dt); bp[:, :N] = b
        lay = K.csr_tiled_layout(data, idx, ptr, M, Kd, dtype=dt)
        outp = torch.empty((M, npad), device="cuda", dtype=dt)
        tl = t(lambda: K.dot_csr_ndarray_tiled(lay, (M, npad), Kd, bp, out=outp))
        line += f", tiled (B padded to {npad} columns, slice not counted) {tl:.3f} ms"
    print(line, flush=True)
