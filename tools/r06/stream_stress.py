"""Repeat the stream kernel many times on config 2's matrix and on odd sizes; every result must equal the first one bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_csr_device  # noqa: E402
from sparse_amd import _kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda")
for (M, Kd, dens) in ((1_000_000, 10_000, 0.01), (200_003, 777, 0.02), (50_000, 3000, 0.3)):
    data, idx, ptr = make_csr_device(M, Kd, dens, seed=1)
    for n_v, dt in ((1, torch.float32), (2, torch.float32), (4, torch.float32), (1, torch.float64), (3, torch.int32)):
        dv = data.to(dt) if dt.is_floating_point else (data * 100).to(dt)
        b = torch.rand((Kd, n_v), device=dev).to(dt) if dt.is_floating_point else torch.randint(-9, 9, (Kd, n_v), device=dev, dtype=dt)
        first = K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b).clone()
        ref = K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b, rowvec=True)
        err = float(((first.double() - ref.double()).abs() / ref.double().abs().clamp_min(1e-30)).max())
        bad = 0
        for _ in range(reps):
            r = K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b)
            bad += int(not torch.equal(r, first))
        torch.cuda.synchronize()
        print(f"M={M} K={Kd} N={n_v} {dt}: {reps} repeats, {bad} differ, rel. diff to the row-vector kernel {err:.1e}", flush=True)
