"""one operand order for counters: dense (128 x 10^6) @ csr (10^6 x 10^4, 3 x 10^7 stored elements), six products"""
import sys
sys.path.insert(0, "/root/repo")
import torch
import sparse_amd as sp
from bench import make_csr_device
from sparse_amd import _settings
_settings.NAN_CHECK = False
M, Kd, N = 1_000_000, 10_000, 128
d, i, p = make_csr_device(M, Kd, 0.003, 5)
a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
bl = torch.rand((N, M), device="cuda")
for _ in range(6):
    c = bl @ a
torch.cuda.synchronize()
