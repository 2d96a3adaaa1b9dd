"""Config 4 (mask 1e5 x 1e5, 1e7 samples, K = 256): the sampled SDDMM kernel in the mask's row-major order against the
column-panel order (spamd_sddmm_panels) at several panel widths.  Prints ms per product and whether the results are
bit-identical."""
import sys, time
import torch
sys.path.insert(0, ".")
import sparse_amd
from sparse_amd import _kernels as K

dev = torch.device("cuda:0")
M = N = 100_000
nnz = 10_000_000
Kd = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev).manual_seed(0)
lin = torch.randperm(M * N // 64, device=dev, generator=g)[:nnz].to(torch.int64) * 64 + torch.randint(0, 64, (nnz,), device=dev, generator=g)
lin = torch.sort(lin).values
coords = torch.stack([lin // N, lin % N]).to(torch.int32)
s = torch.rand(nnz, device=dev, generator=g)


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


for dt in (torch.bfloat16, torch.float32):
    a = torch.rand(M, Kd, device=dev, generator=g).to(dt)
    bt = torch.rand(N, Kd, device=dev, generator=g).to(dt)
    t0, ref = timeit(lambda: K.sddmm_coo(coords, s, a, bt))
    print(f"{dt} K={Kd} row-major order: {t0:.3f} ms", flush=True)
    for width in (3072, 4096, 6144, 6250, 8192, 12500):
        line = f"   panels of {width:6d} Bt rows ({width * Kd * a.element_size() / 2**20:5.2f} MiB):"
        for xcd in (False, True):
            plan = K.sddmm_panels(coords, (M, N), width, xcd=xcd); torch.cuda.synchronize()
            t1, out = timeit(lambda: K.sddmm_coo(coords, s, a, bt, panels=plan))
            line += f"  {'XCD-private' if xcd else 'shared'}: {t1:.3f} ms{'' if torch.equal(ref, out) else ' DIFFERS'}"
        print(line, flush=True)
