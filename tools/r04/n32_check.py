"""Per-call times of the narrow-result product (config-2 matrix x 10000x32) through a @ b."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
import sparse_amd as sp
from sparse_amd import _kernels as K, _ffi

M, Kd = 1_000_000, 10000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
b32 = torch.rand((Kd, 32), device="cuda")
a = sp.GCXS((data, idx, ptr), shape=(M, Kd), compressed_axes=(0,))
for i in range(12):
    torch.cuda.synchronize(); c0 = _ffi.CALLS; t0 = time.perf_counter()
    r = a @ b32
    torch.cuda.synchronize(); print(i, f"{(time.perf_counter() - t0) * 1e3:.3f} ms", "abi calls", _ffi.CALLS - c0, flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): r = a @ b32
e1.record(); torch.cuda.synchronize(); print("events", e0.elapsed_time(e1) / 10)
