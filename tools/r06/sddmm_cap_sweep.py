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
for dt, Kd in ((torch.bfloat16, 384), (torch.float32, 192), (torch.bfloat16, 256), (torch.bfloat16, 128)):
    a = torch.rand(M, Kd, device=dev, generator=g).to(dt); bt = torch.rand(N, Kd, device=dev, generator=g).to(dt)
    plan = K.sddmm_panels(coords, (M, N), K.sddmm_panel_width(bt))
    line = []
    for ch in (0, 48, 52, 56, 58, 60, 62, 64, 66, 72, 80, 96, 128):
        plan.chunk = ch
        t1, got = timeit(lambda: K.sddmm_coo(coords, s, a, bt, panels=plan))
        line.append(f"cap{ch}: {t1:.3f}")
    print(str(dt)[6:], Kd, "width", plan.width, " ".join(line), flush=True)
