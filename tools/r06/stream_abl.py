"""Ablation timings of the stream kernel (variants built with -DSPAMD_STREAM_ABLATE=n; results are wrong by design)."""
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
out = []
for n_v, dt in ((1, torch.float32), (4, torch.float32), (1, torch.float64)):
    dv = data.to(dt)
    b = torch.rand((Kd, n_v), device=dev, dtype=dt)
    ms, _ = timed(lambda: K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b), reps=10)
    out.append(f"N={n_v} {str(dt)[6:]}: {ms:.4f}")
print(os.environ.get("SPAMD_LIB", "default")[-12:], " | ".join(out))
