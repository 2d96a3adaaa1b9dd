"""the matmul family beyond `csr @ dense`: ms per call (steady) and GB/s of algorithmic bytes, to spot operand orders / formats
that fall off the fast paths"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from bench import make_csr_device
from sparse_amd import _settings
_settings.NAN_CHECK = False
def t(f, reps=4):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
M, Kd, N = 1_000_000, 10_000, 128
d, i, p = make_csr_device(M, Kd, 0.003, 5)          # 3e7 stored elements
a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
acsc = a.change_compressed_axes((1,))
acoo = a.asformat("coo")
b = torch.rand((Kd, N), device="cuda"); bl = torch.rand((N, M), device="cuda"); bm = torch.rand((M, N), device="cuda")
cases = [("csr @ dense", lambda: a @ b), ("csc @ dense", lambda: acsc @ b), ("coo @ dense", lambda: acoo @ b),
         ("dense(N x M) @ csr", lambda: bl @ a), ("dense(N x M) @ csc", lambda: bl @ acsc), ("dense(N x M) @ coo", lambda: bl @ acoo),
         ("csr.T @ dense(M x N)", lambda: a.T @ bm), ("csc.T @ dense(M x N)", lambda: acsc.T @ bm), ("coo.T @ dense(M x N)", lambda: acoo.T @ bm),
         ("tensordot(csr, dense, axes=([0],[0]))", lambda: sp.tensordot(a, bm, axes=([0], [0]))), ("dot(csr, dense vec)", lambda: sp.dot(a, b[:, 0].contiguous())),
         ("dense vec @ csr", lambda: bl[0].contiguous() @ a), ("matmul csr, dense numpy-like 3-D? skip", None)]
for name, f in cases:
    if f is None: continue
    try:
        print(f"{name:42s} {t(f):9.3f} ms", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:42s} {type(e).__name__}: {str(e)[:100]}", flush=True)
# 3-D
x = sp.random((256, 256, 256), density=0.01, random_state=1)
w = torch.rand((256, 256), device="cuda", dtype=torch.float64)
for name, f in (("tensordot 3-D axes=1 (config 3 style)", lambda: sp.tensordot(x, w, axes=1)), ("tensordot 3-D axes=([0],[0])", lambda: sp.tensordot(x, w, axes=([0], [0]))),
                ("tensordot 3-D axes=([1],[0])", lambda: sp.tensordot(x, w, axes=([1], [0]))), ("matmul 3-D @ 2-D", lambda: sp.matmul(x, w)),
                ("matmul 3-D @ 3-D sparse", lambda: sp.matmul(x, x)), ("einsum ijk,kl->ijl", lambda: sp.einsum("ijk,kl->ijl", x, w)),
                ("einsum ijk,jl->ilk", lambda: sp.einsum("ijk,jl->ilk", x, w))):
    try:
        print(f"{name:42s} {t(f):9.3f} ms", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:42s} {type(e).__name__}: {str(e)[:100]}", flush=True)
