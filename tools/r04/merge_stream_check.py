"""Single-pass union (merge.hip MODE 2, inputs above the fused form's size) against the two-pass form (count + fill): bit-identical
keys and values over dtypes, functions, overlap patterns and sizes around the tile and grid counts; then timings."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_amd import _umath as U

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(5)

def keys(n, span):
    k = torch.unique(torch.randint(0, span, (int(n * 1.2),), generator=g, device=dev))
    return k[torch.randperm(k.numel(), device=dev, generator=g)[:n]].sort().values if k.numel() > n else k

def run(name, ka, va, kb, vb, fo):
    U.MERGE_SINGLE_PASS = True
    k1, v1 = U.merge_union(name, ka, va, kb, vb, va.new_zeros(()).cpu().numpy()[()], va.new_zeros(()).cpu().numpy()[()], fo)
    U.MERGE_SINGLE_PASS = False
    k2, v2 = U.merge_union(name, ka, va, kb, vb, va.new_zeros(()).cpu().numpy()[()], va.new_zeros(()).cpu().numpy()[()], fo)
    U.MERGE_SINGLE_PASS = True
    same = torch.equal(k1, k2) and torch.equal(v1.view(torch.uint8) if v1.dtype == torch.bool else v1.contiguous().view(torch.uint8),
                                               v2.view(torch.uint8) if v2.dtype == torch.bool else v2.contiguous().view(torch.uint8))
    return same, k1.numel()

bad = 0
cases = []
for dt in (torch.float64, torch.float32, torch.int64, torch.int32):
    for (na, nb, span) in ((5_000_000, 5_000_000, 40_000_000), (9_000_000, 3_000_000, 12_500_000), (8_388_609, 0, 10_000_000),
                           (2_100_000 * 4, 2_100_000 * 4, 2_100_000 * 4 + 7), (12_345_678, 9_876_543, 10 ** 9)):
        ka, kb = keys(na, span), keys(nb, span)
        if dt.is_floating_point:
            va, vb = torch.randn(ka.numel(), device=dev, dtype=dt, generator=g), torch.randn(kb.numel(), device=dev, dtype=dt, generator=g)
        else:
            va = torch.randint(-5, 6, (ka.numel(),), device=dev, dtype=dt, generator=g)
            vb = torch.randint(-5, 6, (kb.numel(),), device=dev, dtype=dt, generator=g)
        for name in ("add", "multiply", "greater"):
            fo = np.zeros((), dtype=np.bool_ if name == "greater" else {torch.float64: np.float64, torch.float32: np.float32,
                                                                        torch.int64: np.int64, torch.int32: np.int32}[dt])[()]
            ok, n = run(name, ka, va, kb, vb, fo)
            bad += not ok
            print(f"{str(dt):14s} {name:9s} na={ka.numel():9d} nb={kb.numel():9d} out={n:9d} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
# identical keys on both sides, 2^20 tiles' worth; and timing
ka = torch.arange(0, 10 ** 8, device=dev) * 3
kb = ka + (torch.arange(0, 10 ** 8, device=dev) % 2)
va = torch.randn(10 ** 8, device=dev, dtype=torch.float64); vb = torch.randn(10 ** 8, device=dev, dtype=torch.float64)
for name in ("add", "multiply"):
    z = np.float64(0)
    for _ in range(2): U.merge_union(name, ka, va, kb, vb, z, z, z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = U.merge_union(name, ka, va, kb, vb, z, z, z)
    e1.record(); torch.cuda.synchronize()
    print(name, "1e8 + 1e8 f64:", e0.elapsed_time(e1) / 5, "ms", r[0].numel())
sys.exit(1 if bad else 0)
