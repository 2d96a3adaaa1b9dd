"""CSR x dense whose RESULT has more than 2^31 elements (17 M rows x 128 columns = 2.2 x 10^9 floats, 8.7 GB): the executor,
the row-group kernel and the stream kernel (N = 4: 6.8 x 10^7... below 2^31; N = 128 is the point) against a float64
evaluation of sampled rows at the head, the middle and the very end of the matrix"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sparse_amd as sp
from bench import make_csr_device
from sparse_amd import _kernels as K, _settings
_settings.NAN_CHECK = False
M, Kd, N = 17_000_000, 1000, 128
d, i, p = make_csr_device(M, Kd, 0.003, 9)
print("nnz", d.numel(), flush=True)
b = torch.rand((Kd, N), device="cuda") - 0.5
a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
def check(c, name):
    rows = np.concatenate([np.arange(0, 5), np.arange(M // 2 - 2, M // 2 + 3), np.arange(16_777_214, 16_777_219), np.arange(M - 5, M)])
    pc = p.cpu().numpy(); ok = True
    for r in rows:
        lo, hi = int(pc[r]), int(pc[r + 1])
        want = (d[lo:hi].double()[:, None] * b[i[lo:hi].long()].double()).sum(0)
        got = c[r].double()
        if not torch.allclose(got, want, rtol=1e-5, atol=1e-6):
            ok = False
            print(name, "row", r, "differs: max abs", float((got - want).abs().max()), flush=True)
    print(name, "ok" if ok else "WRONG", tuple(c.shape), flush=True)
c = a @ b
print("route:", "tiled" if getattr(a, "_tiled_layouts", None) else "general", flush=True)
check(c, "a @ b")
del c
c2 = K.dot_csr_ndarray((M, N), d, i, p, b, keep_order=True)
check(c2, "row-group kernel")
del c2
lay = K.csr_tiled_layout(d, i, p, M, Kd)
c3 = K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b)
check(c3, "executor")
