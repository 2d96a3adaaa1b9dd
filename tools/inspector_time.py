"""One-pass inspector of the tiled SpMM executor at config 2 (10^6 x 10^4 @ 1 %): ms per layout build (HIP events,
allocation included), fp32 and fp64, int32 and int64 indices."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd = 1_000_000, 10_000
for dt in (torch.float32, torch.float64):
    data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234, dtype=dt)
    for it in (torch.int32, torch.int64):
        i2, p2 = idx.to(it), ptr.to(it)
        for _ in range(2):
            layout = K.csr_tiled_layout(data, i2, p2, M, Kd); del layout
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            layout = K.csr_tiled_layout(data, i2, p2, M, Kd); del layout
        e1.record(); torch.cuda.synchronize()
        print(f"inspector {dt} {it}: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
