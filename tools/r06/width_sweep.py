"""CSR x dense at config 2's matrix for every result width a caller may ask for: ms per product through `a @ b` (steady
state: cached layouts), the kernel family that took it, algorithmic bytes / time against 8 TB/s."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import sparse_amd as sp
from bench import make_csr_device, dev_time
from sparse_amd import _settings
_settings.NAN_CHECK = False
M, Kd = 1_000_000, 10_000
dts = (torch.float32, torch.float64) if "f64" in sys.argv else (torch.float32,)
for dt in dts:
    d, i, p = make_csr_device(M, Kd, 0.01, 1234, dtype=dt)
    a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
    for N in (1, 2, 3, 4, 5, 6, 8, 9, 12, 13, 16, 32, 64, 128, 256):
        b = torch.rand((Kd, N), device="cuda", dtype=dt)
        for _ in range(3):
            c = a @ b
        ms = dev_time(lambda: a @ b, 10)
        byt = d.numel() * (d.element_size() + 4) + (M + 1) * 4 + Kd * N * d.element_size() + M * N * d.element_size()
        print(f"{str(dt)[6:]} N={N:4d}: {ms:7.3f} ms   {byt / ms / 1e6 / 8000 * 100:5.1f} % of HBM peak   {2 * d.numel() * N / ms / 1e6:8.1f} GFLOP/s", flush=True)
        del b, c
