"""Executor time at config 2 with the one-pass inspector's layout (row groups at closed-form offsets, tiles + 1 offsets per
group) against the two-pass builder's (one running offsets array): same kernel, same lists."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234)
b = torch.rand((Kd, N), device="cuda")
out = torch.empty((M, N), device="cuda")
res = {}
for rep in range(2):
    for one_pass in (True, False):
        K.TILED_ONE_PASS_INSPECTOR = one_pass
        layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
        f = lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        print(f"one_pass={one_pass} group_ends={getattr(layout, 'group_ends', False)}: {e0.elapsed_time(e1) / 30:.4f} ms", flush=True)
        res[one_pass] = out.clone()
        del layout
print("same result:", bool(torch.equal(res[True], res[False])))
