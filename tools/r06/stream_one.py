"""One shape of the stream / row-vector kernel, a few launches (for rocprofv3 passes): python tools/r06/stream_one.py N dtype [rowvec]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_csr_device  # noqa: E402
from sparse_amd import _kernels as K  # noqa: E402

n_v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
rowvec = len(sys.argv) > 3 and sys.argv[3] == "rowvec"
M, Kd = 1_000_000, 10_000
dev = torch.device("cuda")
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
dv = data.to(dt)
b = torch.rand((Kd, n_v), device=dev, dtype=dt)
for _ in range(6):
    K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b, rowvec=rowvec)
torch.cuda.synchronize()
