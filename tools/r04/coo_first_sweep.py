"""First product of a COO operand: inspector + executor against the row-group kernel, by stored elements and result width
(sets _dot's COO first-product bound)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _dot, _kernels as K, _settings
from bench import make_csr_device

def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for dt, N in ((torch.float32, 128), (torch.float32, 512), (torch.float64, 128), (torch.float64, 512)):
    for M, dens in ((100_000, 0.002), (100_000, 0.004), (200_000, 0.004), (400_000, 0.004), (400_000, 0.01)):
        Kd = 10_000
        data, idx, ptr = make_csr_device(M, Kd, dens, seed=7, dtype=dt)
        b = torch.rand((Kd, N), device="cuda", dtype=dt)
        rows_c = K.csr_to_keys(ptr, torch.zeros_like(idx), M, 1).to(idx.dtype)
        coo = sp.COO(torch.stack([rows_c, idx]), data, shape=(M, Kd), has_duplicates=False, sorted=True)
        if not _dot._tiled_eligible(data, b, (M, N), Kd):
            print(f"{str(dt):14s} N={N:4d} nnz={data.numel():9d}: not eligible"); continue
        def first(force):
            _dot.drop_derived(coo); coo.__dict__.pop("_spmm_uses", None)
            old = _dot.COO_TILED_FIRST_NNZ
            _dot.COO_TILED_FIRST_NNZ = 0 if force else 1 << 62
            try: return coo @ b
            finally: _dot.COO_TILED_FIRST_NNZ = old
        ti, tr = t(lambda: first(True)), t(lambda: first(False))
        rb = N * b.element_size()
        print(f"{str(dt):14s} N={N:4d} nnz={data.numel():9d} nnz*row_bytes={data.numel() * rb / 1e9:7.2f}e9: inspector at first {ti:.3f} ms, row-group first {tr:.3f} ms, ratio {tr / ti:.2f}", flush=True)
