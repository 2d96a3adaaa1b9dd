"""Phase timeline of one workgroup of the tiled SpMM executor (SPAMD_TILED_DBG=20): per tile and wave,
s_memtime at loop top (after the barrier), after the tile-DMA issue, after the list loop, after the DMA wait."""
import os, sys, numpy as np, torch
os.environ["SPAMD_TILED_DBG"] = "20"
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1234)
b = torch.rand((Kd, N), device="cuda")
layout = K.csr_tiled_layout(data, idx, ptr, M, Kd)
out = torch.empty((M, N), device="cuda")
for _ in range(3):
    K.dot_csr_ndarray_tiled(layout, (M, N), Kd, b, out=out)
torch.cuda.synchronize()
wg = (-(-M // 512)) // 2
raw = out[wg * 512: wg * 512 + 40].contiguous().view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1)[: 16 * 80 * 4]
t = (raw & 0xffffffff).reshape(16, 80, 4)[:, :79]  # [wave, tile, stamp]
t0 = t[:, :, 0].min(axis=0)                          # first wave past the barrier, per tile
ph = np.diff(t0).astype(np.int64) & 0xffffffff
print("tile period (s_memtime ticks): median", np.median(ph), "mean", ph[1:-1].mean())
d_issue = (t[:, :, 1] - t[:, :, 0]) & 0xffffffff
d_cons = (t[:, :, 2] - t[:, :, 1]) & 0xffffffff
d_wait = (t[:, :, 3] - t[:, :, 2]) & 0xffffffff
skew = (t[:, :, 0] - t0[None, :]) & 0xffffffff
end = (t[:, :, 3] - t0[None, :]) & 0xffffffff
for name, a in (("barrier exit skew", skew), ("DMA issue", d_issue), ("list loop", d_cons), ("touch + DMA wait", d_wait), ("arrive at barrier (from tile start)", end)):
    a = a[:, 2:77]
    print(f"{name:38s} mean {a.mean():8.1f}  median {np.median(a):8.1f}  max-over-waves mean {a.max(axis=0).mean():8.1f}  min-over-waves mean {a.min(axis=0).mean():8.1f}")
nb = (layout[1][wg * 16 * 79: (wg * 16 + 16) * 79 + 1]).cpu().numpy()
print("blocks per list in this WG: mean", np.diff(nb).mean())
