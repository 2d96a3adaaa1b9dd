"""Config 2's matrix times a vector / 2 / 4 columns: the stream form against the row-vector kernel it replaces
(bash tools/gpu_job.sh py tools/r06/stream_time.py [mults])."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_csr_device  # noqa: E402
from bench_paths import timed  # noqa: E402
from sparse_amd import _ffi, _kernels as K  # noqa: E402
from sparse_amd._device import code_of, ptr as p_, stream_ptr  # noqa: E402

M, Kd, nnz = 1_000_000, 10_000, 100_000_000
dev = torch.device("cuda")
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
nnz = int(data.numel())
mults = [int(x, 0) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4]   # flags >> 8: low byte = grid multiple, 0x100 = 1024 threads
for n_v, dt in ((1, torch.float32), (2, torch.float32), (4, torch.float32), (1, torch.float64)):
    dv = data.to(dt)
    b = torch.rand((Kd, n_v), device=dev, dtype=dt)
    es = dv.element_size()
    alg = nnz * (es + 4) + (M + 1) * 4 + Kd * n_v * es + M * n_v * es
    ms_rv, r_rv = timed(lambda: K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b, rowvec=True), reps=10)
    ms_rg, r_rg = timed(lambda: K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b, keep_order=True), reps=3)
    line = f"N={n_v} {dt}: rowvec {ms_rv:.4f} ms ({alg / ms_rv / 8e9 * 100:.1f} %)"
    for mult in mults:
        out = torch.empty((M, n_v), dtype=dt, device=dev)

        def run():
            _ffi.call("spamd_spmm_csr_stream", code_of(dt), code_of(idx.dtype), M, Kd, n_v, p_(dv), p_(idx), p_(ptr), p_(b),
                      n_v, p_(out), n_v, nnz, mult << 8, stream_ptr(dev))
            return out
        ms, r = timed(run, reps=10)
        err = float(((r.double() - r_rg.double()).abs() / r_rg.double().abs().clamp_min(1e-30)).max())
        line += f" | stream {mult:#x} {ms:.4f} ms ({alg / ms / 8e9 * 100:.1f} %) relerr {err:.1e}"
    print(line, flush=True)
