"""natural vs balanced executor layout on the Zipf matrix of config 2's size (and on the uniform one): ms per product"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import make_powerlaw_csr_device, make_csr_device, dev_time
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
b = torch.rand((Kd, N), device="cuda")
for name, (d, i, p) in (("zipf", make_powerlaw_csr_device(M, Kd, 100_000_000, 21)), ("uniform", make_csr_device(M, Kd, 0.01, 1234))):
    ref = K.dot_csr_ndarray((M, N), d, i, p, b)
    for bal in (False, True):
        K.TILED_BALANCE = bal
        for skew in ((1.6,) if not bal else (0.0,)):
            K.TILED_BALANCE_SKEW = skew
            for capmul in ((4,) if not bal else (1, 2, 4, 8)):
                K.TILED_BALANCE_CAPMUL = capmul
                torch.cuda.synchronize(); t0 = time.perf_counter()
                lay = K.csr_tiled_layout(d, i, p, M, Kd); torch.cuda.synchronize()
                t_ins = (time.perf_counter() - t0) * 1e3
                out = K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b)
                same = bool(torch.equal(out, ref))
                ms = dev_time(lambda: K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b, out=out), 10)
                print(name, "balanced" if lay.rowmap is not None else "natural", "capmul", capmul, f"inspector {t_ins:.2f} ms  product {ms:.3f} ms  identical {same}",
                      {k: v for k, v in K.TILED_BALANCE_STATS.items() if k in ("cap", "groups", "skew", "class_rows")} if bal else "", flush=True)
                del lay
    del d, i, p, ref
    torch.cuda.empty_cache()
