"""The matrix-vector product with the reference's DEFAULT dtypes (float64 values, int64 indices) at config 2's size, and fp32 / int64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_csr_device  # noqa: E402
from bench_paths import timed  # noqa: E402
from sparse_amd import _kernels as K  # noqa: E402

M, Kd = 1_000_000, 10_000
dev = torch.device("cuda")
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
nnz = int(data.numel())
for dt, it in ((torch.float64, torch.int64), (torch.float32, torch.int64), (torch.float64, torch.int32)):
    dv, iv, pv = data.to(dt), idx.to(it), ptr.to(it)
    for n_v in (1, 2):
        b = torch.rand((Kd, n_v), device=dev, dtype=dt)
        es, isz = dv.element_size(), iv.element_size()
        alg = nnz * (es + isz) + (M + 1) * isz + Kd * n_v * es + M * n_v * es
        ms, r = timed(lambda: K.dot_csr_ndarray((M, n_v), dv, iv, pv, b), reps=10)
        ms_rv, r2 = timed(lambda: K.dot_csr_ndarray((M, n_v), dv, iv, pv, b, rowvec=True), reps=5)
        print(f"{str(dt)[6:]} / {str(it)[6:]} N={n_v}: stream {ms:.4f} ms ({alg / ms / 8e9 * 100:.1f} %), row-vector kernel {ms_rv:.4f} ms", flush=True)
