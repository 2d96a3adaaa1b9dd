"""Config 4 (mask 1e5 x 1e5, 1e7 samples, K = 256), default panel plan: ms per product, bf16 and fp32."""
import sys, torch
sys.path.insert(0, "/root/repo")
import sparse_amd
from sparse_amd import _kernels as K
dev = torch.device("cuda:0")
M = N = 100_000; nnz = 10_000_000; Kd = 256
g = torch.Generator(device=dev).manual_seed(0)
lin = torch.randperm(M * N // 64, device=dev, generator=g)[:nnz].to(torch.int64) * 64 + torch.randint(0, 64, (nnz,), device=dev, generator=g)
lin = torch.sort(lin).values
coords = torch.stack([lin // N, lin % N]).to(torch.int32)
s = torch.rand(nnz, device=dev, generator=g)
def timeit(f, n=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
out = []
for dt in (torch.bfloat16, torch.float32):
    a = torch.rand(M, Kd, device=dev, generator=g).to(dt); bt = torch.rand(N, Kd, device=dev, generator=g).to(dt)
    t0, ref = timeit(lambda: K.sddmm_coo(coords, s, a, bt))
    plan = K.sddmm_panels(coords, (M, N), K.sddmm_panel_width(bt))
    t1, got = timeit(lambda: K.sddmm_coo(coords, s, a, bt, panels=plan))
    out.append(f"{str(dt)[6:]}: row-major {t0:.3f} ms, panels(width {plan.width}) {t1:.3f} ms, identical {torch.equal(ref, got)}")
print(" | ".join(out))
if len(sys.argv) > 1 and sys.argv[1] == "chunks":
    for dt in (torch.bfloat16, torch.float32):
        a = torch.rand(M, Kd, device=dev, generator=g).to(dt); bt = torch.rand(N, Kd, device=dev, generator=g).to(dt)
        plan = K.sddmm_panels(coords, (M, N), K.sddmm_panel_width(bt))
        ref0 = K.sddmm_coo(coords, s, a, bt)
        line = []
        for ch in (0, 16, 20, 24, 28, 32, 48):
            plan.chunk = ch
            t1, got = timeit(lambda: K.sddmm_coo(coords, s, a, bt, panels=plan))
            line.append(f"cap {ch}: {t1:.3f}{'' if torch.equal(got, ref0) else ' WRONG'}")
        print(str(dt)[6:], " ".join(line))
