"""A/B of first-product and float64 rows between two source trees on ONE box: python ab_first.py <tree root>"""
import sys, os
root = sys.argv[1]
sys.path.insert(0, root)
import torch, sparse_amd as sp
from sparse_amd import _kernels as K, _dot, _settings
sys.path.insert(0, "/root/repo")
from bench import make_csr_device, dev_time
_settings.NAN_CHECK = False
M, Kd, N = 1_000_000, 10_000, 128
d, i, p = make_csr_device(M, Kd, 0.01, 1234)
a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
b = torch.rand((Kd, N), device="cuda")
def first(x, bb):
    _dot.drop_derived(x); x.__dict__.pop("_spmm_uses", None)
    return x @ bb
for _ in range(3): first(a, b)
print(os.path.basename(root.rstrip("/")) or root, "fp32 first product", round(dev_time(lambda: first(a, b), 10), 4), "steady", round(dev_time(lambda: a @ b, 20), 4),
      "inspector", round(dev_time(lambda: K.csr_tiled_layout(d, i, p, M, Kd, defer_check=True), 10), 4), flush=True)
a64 = sp.GCXS((d.double(), i, p), shape=(M, Kd), compressed_axes=(0,)); b64 = b.double()
for _ in range(3): first(a64, b64)
print("   f64 first product", round(dev_time(lambda: first(a64, b64), 5), 4), "steady", round(dev_time(lambda: a64 @ b64, 10), 4), flush=True)
