"""Randomised cross-check of the stream form of CSR x dense (spmm_stream.hip, N <= 4) against the k-ascending row-group kernel:
integers bit for bit, floating point within eps-scaled sum |a_k b_k| (the row-group kernel on the absolute values gives the
bound); uniform and Zipf row lengths, empty matrices, rows longer than a subtile, every dtype / index width / N, odd sizes.
    python tools/fuzz_stream.py [seconds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
from bench import make_csr_device, make_powerlaw_csr_device  # noqa: E402
from sparse_amd import _ffi, _kernels as K  # noqa: E402
from sparse_amd._device import code_of, ptr as p_, stream_ptr  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
dev = torch.device("cuda")
t_end = time.time() + budget
cases = skipped = 0
while time.time() < t_end:
    M = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 33000, 150001, 600000]))
    Kd = int(rng.choice([1, 2, 100, 511, 4096, 9999, 20000, 40000]))
    it = torch.int32 if rng.random() < 0.6 else torch.int64
    dens = float(rng.choice([0.0, 1e-5, 0.001, 0.02, 0.3, 1.0]))
    if M * Kd * dens > 3e6:
        dens = 3e6 / (M * Kd)
    seed = int(rng.integers(1 << 30))
    if rng.random() < 0.35 and M >= 64 and Kd >= 100 and dens > 0:
        data, idx, ptr = make_powerlaw_csr_device(M, Kd, max(int(M * Kd * dens), 1), seed=seed, idx_dtype=it)
    else:
        data, idx, ptr = make_csr_device(M, Kd, dens, seed=seed, idx_dtype=it)
    for _ in range(6):      # several (dtype, N) products per generated matrix
        N = int(rng.integers(1, 5)) if rng.random() < 0.6 else int(rng.integers(5, 14))     # (5 and more: several passes, round 6)
        dt = [torch.float32, torch.float64, torch.int32, torch.int64][int(rng.integers(0, 4))]
        if dt.is_floating_point:
            dv = (data - 0.4).to(dt)
            b = (torch.rand((Kd, N), device=dev, dtype=torch.float64) - 0.5).to(dt)
        else:
            dv = ((data - 0.5) * 200).to(dt)
            b = torch.randint(-90, 90, (Kd, N), device=dev, dtype=dt)
        vc = code_of(dt)
        if not _ffi.lib().spamd_spmm_csr_stream_fits(vc, M, Kd, N, p_(dv), p_(idx)):
            skipped += 1
            continue
        out = torch.full((M, N), 7, dtype=dt, device=dev)
        nnz_arg = int(dv.numel()) if rng.random() < 0.5 else -1
        mult = int(rng.choice([0, 0, 1, 3]))
        _ffi.call("spamd_spmm_csr_stream", vc, code_of(it), M, Kd, N, p_(dv), p_(idx), p_(ptr), p_(b), N, p_(out), N, nnz_arg, mult << 8,
                  stream_ptr(dev))
        ref = K.dot_csr_ndarray((M, N), dv, idx, ptr, b, keep_order=True)
        if dt.is_floating_point:
            bound = K.dot_csr_ndarray((M, N), dv.abs(), idx, ptr, b.abs(), keep_order=True).double()
            eps = 1.2e-7 if dt == torch.float32 else 2.3e-16
            bad = ((out.double() - ref.double()).abs() > 8 * eps * bound + 1e-300)
            ok = not bool(bad.any())
        else:
            ok = torch.equal(out, ref)
        if not ok:
            print(f"MISMATCH M={M} K={Kd} N={N} {dt} {it} dens={dens} seed={seed} nnz={int(dv.numel())}", flush=True)
            sys.exit(1)
        cases += 1
print(f"fuzz_stream: {cases} cases agree with the row-group kernel ({skipped} shapes declined by spamd_spmm_csr_stream_fits)")
