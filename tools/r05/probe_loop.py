"""why does a 2000-call loop of the 1000^3 dense product take 80 us per call when a 200-call loop takes 19?"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, sparse_amd as sp
x = sp.random((1000, 1000), density=0.001, random_state=1, format="gcxs", compressed_axes=(0,))
t = torch.rand((1000, 1000), device="cuda", dtype=torch.float64)
for _ in range(50): x @ t
torch.cuda.synchronize()
for reps in (100, 200, 400, 800, 1600, 3200):
    t0 = time.perf_counter()
    for _ in range(reps): x @ t
    e = time.perf_counter() - t0
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    print(reps, f"enqueue {e/reps*1e6:.1f} us  wall {w/reps*1e6:.1f} us", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): x @ t
e1.record(); torch.cuda.synchronize()
print("device time per product", e0.elapsed_time(e1) / 200 * 1e3, "us")
from sparse_amd import _settings
_settings.NAN_CHECK = False
x2 = sp.random((1000, 1000), density=0.001, random_state=1, format="gcxs", compressed_axes=(0,))
for _ in range(5): x2 @ t
torch.cuda.synchronize()
e0.record()
for _ in range(200): x2 @ t
e1.record(); torch.cuda.synchronize()
print("device time per product without the NaN scan", e0.elapsed_time(e1) / 200 * 1e3, "us")
