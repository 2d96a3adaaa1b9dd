"""CSR x dense with more than 2^31 STORED elements (3 x 10^6 rows x 750 per row = 2.25 x 10^9; int64 pointers): the stream
kernel (N = 1, 4), the row-group kernel and whatever `a @ b` picks for N = 128, on sampled rows"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from sparse_amd import _kernels as K, _settings
_settings.NAN_CHECK = False
dev = torch.device("cuda:0")
M, Kd, per = 3_000_000, 10_000, 750
idx = ((torch.arange(per, device=dev, dtype=torch.int32) * 13)[None, :] + (torch.arange(M, device=dev, dtype=torch.int32) % 17)[:, None]).reshape(-1)
ptr = torch.arange(M + 1, device=dev, dtype=torch.int64) * per
data = torch.rand(M * per, device=dev) - 0.5
print("nnz", idx.numel(), idx.numel() > 2 ** 31, flush=True)
rows = np.concatenate([np.arange(0, 3), np.arange(2 ** 31 // per - 2, 2 ** 31 // per + 3), np.arange(M - 3, M)])
def check(c, b, name):
    ok = True
    for r in rows:
        lo, hi = r * per, (r + 1) * per
        want = (data[lo:hi].double()[:, None] * b[idx[lo:hi].long()].double()).sum(0)
        if not torch.allclose(c[int(r)].double(), want, rtol=1e-4, atol=1e-5):
            ok = False
            print(name, "row", r, "max abs", float((c[int(r)].double() - want).abs().max()), flush=True)
    print(name, "ok" if ok else "WRONG", flush=True)
for N in (1, 4, 8):
    b = torch.rand((Kd, N), device=dev) - 0.5
    t0 = time.perf_counter(); c = K.dot_csr_ndarray((M, N), data, idx, ptr, b); torch.cuda.synchronize()
    print(f"N={N}: {(time.perf_counter() - t0) * 1e3:.1f} ms (first call, conversions included)", flush=True)
    check(c, b, f"dot_csr_ndarray N={N} (stream passes {K.stream_passes(M, Kd, N, torch.float32, data, idx.long())})")
    check(K.dot_csr_ndarray((M, N), data, idx, ptr, b, keep_order=True), b, f"row-group N={N}")
b = torch.rand((Kd, 128), device=dev) - 0.5
a = sp.GCXS((data, idx, ptr), shape=(M, Kd), compressed_axes=(0,))
c = a @ b
print("route N=128:", "tiled" if getattr(a, "_tiled_layouts", None) else "general", flush=True)
check(c, b, "a @ b N=128")
