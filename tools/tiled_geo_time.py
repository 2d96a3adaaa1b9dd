"""Executor time at config 2 for whatever geometry the loaded library was built with (layout by the key-sort recipe, which
works for any geometry), bit-compared with the row-group kernel."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234)
b = torch.rand((Kd, N), device="cuda")
layout = K.csr_tiled_layout(data, idx, ptr, M, Kd, force_sort="--direct" not in sys.argv)
out = torch.empty((M, N), device="cuda")
f = lambda: K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)
for _ in range(60): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
c = K.dot_csr_ndarray((M, N), data, idx, ptr, b)
print(f"geometry {K.tiled_params()[:4]}: {e0.elapsed_time(e1)/20:.3f} ms, blocks {layout[0].numel()//16}, bit-identical to row-group: {torch.equal(out, c)}")
if "--empty" in sys.argv:
    # every list empty (all offsets 0): what is left is the tile DMA, the per-tile barrier and the list heads
    empty = K.TiledLayout(layout[0], torch.zeros_like(layout[1]), layout[2], 1.0, group_ends=getattr(layout, "group_ends", False))
    g = lambda: K.dot_csr_ndarray_tiled(empty, (M, N), Kd, b, out=out)
    for _ in range(20): g()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): g()
    e1.record(); torch.cuda.synchronize()
    print(f"all lists empty (DMA + barriers + heads only): {e0.elapsed_time(e1)/20:.3f} ms")
