import sys
root = sys.argv[1]
sys.path.insert(0, root)
import torch
from sparse_amd import _kernels as K
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
M, Kd = 1_000_000, 10_000
d, i, p = make_csr_device(M, Kd, 0.01, 1234)
for _ in range(12): K.csr_tiled_layout(d, i, p, M, Kd, defer_check=True)
torch.cuda.synchronize()
