"""CSR x dense (N = 128) over shapes and value types around config 2: ms per product through `a @ b` (first and steady),
normalised to ns per stored element, to spot cliffs in the dispatch (inspector limits, policies, dtypes without an executor)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import sparse_amd as sp
from bench import make_csr_device, dev_time
from sparse_amd import _settings, _dot
_settings.NAN_CHECK = False
N = 128
def run(M, Kd, dens, dt, note=""):
    vt = dt if dt.is_floating_point and dt not in (torch.float16, torch.bfloat16) else torch.float32
    d, i, p = make_csr_device(M, Kd, dens, 77, dtype=vt)
    if dt != vt:
        d = (d * 50).to(dt) if not dt.is_floating_point else d.to(dt)
    a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
    b = torch.rand((Kd, N), device="cuda", dtype=torch.float32)
    b = (b * 9).to(dt) if not dt.is_floating_point else b.to(dt)
    try:
        import time
        torch.cuda.synchronize(); t = time.perf_counter(); c = a @ b; torch.cuda.synchronize(); first = (time.perf_counter() - t) * 1e3
        for _ in range(2): c = a @ b
        ms = dev_time(lambda: a @ b, 5)
        route = "tiled" if getattr(a, "_tiled_layouts", None) else "general"
        print(f"M={M:8d} K={Kd:7d} dens={dens:.4g} {str(dt)[6:]:8s} nnz={d.numel():10d}: steady {ms:8.3f} ms  ({ms * 1e6 / max(d.numel(), 1):6.2f} ns/nnz)  first {first:8.2f} ms  {route} {note}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"M={M} K={Kd} dens={dens} {dt}: {type(e).__name__} {str(e)[:80]}", flush=True)
    del a, b
    torch.cuda.empty_cache()
for Kd in (1000, 5000, 20_000, 40_000, 41_000, 100_000, 1_000_000):
    run(1_000_000, Kd, 3e7 / (1_000_000 * Kd), torch.float32, "K sweep, 3e7 nnz")
for M in (10_000, 50_000, 200_000, 4_000_000):
    run(M, 10_000, 0.01, torch.float32, "M sweep")
for dens in (0.0001, 0.001, 0.003, 0.03, 0.1):
    run(300_000, 10_000, dens, torch.float32, "density sweep")
for dt in (torch.float64, torch.int32, torch.int64, torch.float16, torch.bfloat16, torch.int16):
    run(300_000, 10_000, 0.01, dt, "dtype sweep")
