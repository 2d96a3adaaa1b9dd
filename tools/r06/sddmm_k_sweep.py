"""SDDMM at config 4's mask (10^5 x 10^5, 10^7 samples) for inner dimensions other than 256: ms per product through the
product path (row-major order and the cached column-panel plan), GB/s of rows gathered (samples x 2 rows x K x bytes)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from sparse_amd import _kernels as K
dev = torch.device("cuda:0")
M = N = 100_000; nnz = 10_000_000
g = torch.Generator(device=dev).manual_seed(0)
lin = torch.randperm(M * N // 64, device=dev, generator=g)[:nnz].to(torch.int64) * 64 + torch.randint(0, 64, (nnz,), device=dev, generator=g)
lin = torch.sort(lin).values
coords = torch.stack([lin // N, lin % N]).to(torch.int32)
s = torch.rand(nnz, device=dev, generator=g)
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
for dt in (torch.bfloat16, torch.float32, torch.float64):
    for Kd in (16, 32, 64, 96, 100, 128, 192, 256, 384, 512, 1024):
        a = torch.rand(M, Kd, device=dev, generator=g).to(dt); bt = torch.rand(N, Kd, device=dev, generator=g).to(dt)
        t0, ref = timeit(lambda: K.sddmm_coo(coords, s, a, bt))
        try:
            plan = K.sddmm_panels(coords, (M, N), K.sddmm_panel_width(bt))
            t1, got = timeit(lambda: K.sddmm_coo(coords, s, a, bt, panels=plan))
            same = torch.equal(ref, got)
        except Exception as e:      # noqa: BLE001
            t1, same = float("nan"), str(e)[:60]
        print(f"{str(dt)[6:]:9s} K={Kd:5d}: row-major {t0:7.3f} ms   panels {t1:7.3f} ms   identical {same}   "
              f"gather rate (panels) {nnz * 2 * Kd * a.element_size() / t1 / 1e6:7.0f} GB/s", flush=True)
        del a, bt
