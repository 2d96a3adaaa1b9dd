import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd = 1_000_000, 10_000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234)
for force in (False, True, False, True):
    torch.cuda.synchronize(); t = time.perf_counter()
    layout = K.csr_tiled_layout(data, idx, ptr, M, Kd, force_sort=force)
    torch.cuda.synchronize(); print(f"inspector (force_sort={force}): {(time.perf_counter()-t)*1e3:.2f} ms")
    del layout
