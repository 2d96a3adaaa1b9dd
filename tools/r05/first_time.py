"""what a uniform operand's FIRST product costs (config 2): inspector alone and inspector + executor through a @ b, with and
without the balanced-layout machinery"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, sparse_amd as sp
from bench import make_csr_device, dev_time
from sparse_amd import _kernels as K, _dot
M, Kd = 1_000_000, 10_000
d, i, p = make_csr_device(M, Kd, 0.01, 1234)
a = sp.GCXS((d, i, p), shape=(M, Kd), compressed_axes=(0,))
b = torch.rand((Kd, 128), device="cuda")
def first():
    _dot.drop_derived(a)
    return a @ b
for bal in (False, True, False, True):
    K.TILED_BALANCE = bal
    for _ in range(3): K.csr_tiled_layout(d, i, p, M, Kd, defer_check=True)
    ms = dev_time(lambda: K.csr_tiled_layout(d, i, p, M, Kd, defer_check=True), 10)
    for _ in range(3): first()
    msf = dev_time(first, 10)
    lay = a._tiled_layouts[torch.float32]
    mse = dev_time(lambda: K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, b), 10)
    print("balance machinery", bal, f"inspector {ms:.3f} ms, first product {msf:.3f} ms, executor {mse:.3f} ms", flush=True)
