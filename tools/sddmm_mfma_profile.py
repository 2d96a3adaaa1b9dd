"""The clustered-mask SDDMM of bench_paths (A9_mfma_clustered) alone, for rocprofv3 counter passes:
   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/sddmm_mfma_profile.py"""
import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K

Ms, nnz = 100_000, 10_000_000
dev = torch.device("cuda")
a = (torch.rand((Ms, 256), device=dev) - 0.5).to(torch.bfloat16)
bt = (torch.rand((Ms, 256), device=dev) - 0.5).to(torch.bfloat16)
rng = np.random.default_rng(1)
nt = int(0.7 * nnz) // 512
tiles = rng.choice((Ms // 32) ** 2, nt, replace=False)
pos = np.argsort(rng.random((nt, 1024)), axis=1)[:, :512]
r = (tiles // (Ms // 32))[:, None] * 32 + pos // 32
c = (tiles % (Ms // 32))[:, None] * 32 + pos % 32
lin = np.unique(np.concatenate([(r.astype(np.int64) * Ms + c).ravel(), rng.choice(Ms * Ms, nnz - nt * 512, replace=False)]))
mask = sp.COO(np.stack([lin // Ms, lin % Ms]).astype(np.int32), rng.random(lin.size).astype(np.float32), shape=(Ms, Ms))
plan = K.sddmm_plan(mask.coords, mask.shape)
w = K.sddmm_panel_width(bt)
restp = K.sddmm_panels(mask.coords, mask.shape, w, subset=plan.rest)
allp = K.sddmm_panels(mask.coords, mask.shape, w)
for _ in range(5):
    K.sddmm_coo_mfma(plan, mask.coords, mask.shape, mask.data, a, bt, force=True, rest_panels=restp)
for _ in range(5):
    K.sddmm_coo(mask.coords, mask.data, a, bt, panels=allp)
torch.cuda.synchronize()
print("dense tiles", int(plan.tiles.numel()), "samples in them", plan.n_dense_samples)
