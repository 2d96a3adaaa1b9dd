"""Per-wave timestamps of the stream kernel (a -DSPAMD_STREAM_PROF variant: SPAMD_LIB=.../libsparse_amd_prof.so)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_csr_device  # noqa: E402
from sparse_amd import _ffi, _kernels as K  # noqa: E402

M, Kd = 1_000_000, 10_000
dev = torch.device("cuda")
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
for n_v, dt in ((1, torch.float32), (4, torch.float32), (1, torch.float64)):
    dv = data.to(dt)
    b = torch.rand((Kd, n_v), device=dev, dtype=dt)
    for _ in range(3):
        K.dot_csr_ndarray((M, n_v), dv, idx, ptr, b)
    torch.cuda.synchronize()
    nw = 4096
    buf = np.zeros(nw * 4, dtype=np.uint64)
    fn = _ffi.lib().spamd_stream_prof_read
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fn(buf.ctypes.data, nw * 4) == 0
    t = buf.reshape(nw, 4).astype(np.float64) / 100.0   # us (100 MHz)
    t0 = t[:, 0].min()
    t -= t0
    q = lambda x: " ".join(f"{v:7.2f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))
    print(f"N={n_v} {dt}: percentiles 0/10/50/90/100 over {nw} waves (us since the first wave's entry)")
    print("  entry            ", q(t[:, 0]))
    print("  search done      ", q(t[:, 1]), "  (search: ", q(t[:, 1] - t[:, 0]), ")")
    print("  B copied, barrier", q(t[:, 2]), "  (+", q(t[:, 2] - t[:, 1]), ")")
    print("  piece done       ", q(t[:, 3]), "  (stream: ", q(t[:, 3] - t[:, 2]), ")")
    nwv = 16
    blk = np.arange(nw) // nwv
    end = t[:, 3]
    print("  end by XCD (block % 8):   ", " ".join(f"{np.median(end[blk % 8 == x]):7.1f}" for x in range(8)))
    print("  end by wave slot in block:", " ".join(f"{np.median(end[np.arange(nw) % nwv == x]):6.1f}" for x in range(nwv)))
    per_blk = end.reshape(-1, nwv)
    print("  per-block max-min of end: ", q(per_blk.max(1) - per_blk.min(1)), " block medians:", q(np.median(per_blk, 1)))
    if not os.environ.get("PHASES"):
        continue
    ph = np.zeros(nw * 8, dtype=np.uint64)
    fn2 = _ffi.lib().spamd_stream_phase_read
    fn2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fn2(ph.ctypes.data, nw * 8) == 0
    ph = ph.reshape(nw, 8).astype(np.float64)
    names = ["ring wait", "phase A+mask", "gather+lane", "scan", "phase B", "-", "-", "-"]
    for slot in (0, 4, 8, 12):
        sel = np.arange(nw) % nwv == slot
        tot = ph[sel].sum(1).mean()
        print(f"  slot {slot:2d}: memtime ticks per wave {tot:9.0f}: " + ", ".join(f"{names[k]} {ph[sel][:, k].mean() / tot * 100:4.1f}%" for k in range(5)))
