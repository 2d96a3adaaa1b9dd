"""Crossover between the sampled SDDMM kernel and the matrix-core tile kernel (csrc/sddmm_mfma.hip) as a function of how
populated the mask's 32 x 32 tiles are.  Masks: `nnz` samples spread over randomly placed tiles filled at density d;
A, Bt bf16 (M = N = 65536, K = 256).  Prints ns per sample for both kernels and the MFMA/sampled ratio."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K

M = N = 65536
Kd = 256
NNZ = 4_000_000
dev = torch.device("cuda")
at = (torch.rand((M, Kd), device=dev) - 0.5).to(torch.bfloat16)
bt = (torch.rand((N, Kd), device=dev) - 0.5).to(torch.bfloat16)
rng = np.random.default_rng(0)


def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{'samples/tile':>12} {'tiles':>9} {'sampled ns/sample':>18} {'panel order':>12} {'mfma ns/sample':>15} {'mfma/sampled':>13} {'mfma/panel':>11}")
for per_tile in (4, 8, 12, 16, 24, 32, 48, 64, 128, 256, 512, 1024):
    nt = NNZ // per_tile
    tiles = rng.choice((M // 32) * (N // 32), nt, replace=False)
    pos = np.argsort(rng.random((nt, 1024)), axis=1)[:, :per_tile] if per_tile < 1024 else np.tile(np.arange(1024), (nt, 1))
    r = (tiles // (N // 32))[:, None] * 32 + pos // 32
    c = (tiles % (N // 32))[:, None] * 32 + pos % 32
    lin = np.sort((r.astype(np.int64) * N + c).ravel())
    coords = torch.from_numpy(np.stack([lin // N, lin % N]).astype(np.int32)).to(dev)
    sval = torch.rand(lin.size, device=dev)
    plan = K.sddmm_plan(coords, (M, N), threshold=1)          # every tile on the matrix cores
    out = torch.empty(lin.size, dtype=torch.float32, device=dev)
    t_m = timeit(lambda: K.sddmm_coo_mfma(plan, coords, (M, N), sval, at, bt, out=out, force=True))
    t_s = timeit(lambda: K.sddmm_coo(coords, sval, at, bt))
    panels = K.sddmm_panels(coords, (M, N), K.sddmm_panel_width(bt))
    t_p = timeit(lambda: K.sddmm_coo(coords, sval, at, bt, panels=panels))
    print(f"{per_tile:12d} {nt:9d} {t_s * 1e6 / lin.size:18.3f} {t_p * 1e6 / lin.size:12.3f} {t_m * 1e6 / lin.size:15.3f} "
          f"{t_m / t_s:13.2f} {t_m / t_p:11.2f}", flush=True)

# BASELINE config 4 through the product entry point: the dispatcher must leave it on the sampled kernel
M4 = N4 = 100_000
s4 = sp.random((M4, N4), nnz=10_000_000, random_state=1, dtype=np.float32, idx_dtype=np.int32)
a4 = (torch.rand((M4, Kd), device=dev) - 0.5).to(torch.bfloat16)
b4 = (torch.rand((N4, Kd), device=dev) - 0.5).to(torch.bfloat16)
t0 = time.time(); sp.sddmm(s4, a4, bt=b4); torch.cuda.synchronize(); first = (time.time() - t0) * 1e3
p = s4._sddmm_plan[("tiles", K.SDDMM_TILE_THRESHOLD)]
t_auto = timeit(lambda: sp.sddmm(s4, a4, bt=b4))
t_samp = timeit(lambda: K.sddmm_coo(s4.coords, s4.data, a4, b4))
t_pan = timeit(lambda: K.sddmm_coo(s4.coords, s4.data, a4, b4, panels=s4._sddmm_plan[("panels", "all", K.sddmm_panel_width(b4))]))
print(f"config 4 (uniform 0.1 %): dense tiles {p.tiles.numel()}, samples in them {p.n_dense_samples}; sparse_amd.sddmm {t_auto:.3f} ms "
      f"(kernel + pruning + container), sampled kernel alone {t_samp:.3f} ms row-major / {t_pan:.3f} ms panel order, "
      f"first call incl. plans {first:.1f} ms")
# a block-clustered mask of the same size: 10^7 samples, 70 % of them in tiles filled at 50 %
nt = 7_000_000 // 512
tiles = rng.choice((M4 // 32) * (N4 // 32), nt, replace=False)
pos = np.argsort(rng.random((nt, 1024)), axis=1)[:, :512]
r = (tiles // (N4 // 32))[:, None] * 32 + pos // 32
c = (tiles % (N4 // 32))[:, None] * 32 + pos % 32
lin = np.unique(np.concatenate([(r.astype(np.int64) * N4 + c).ravel(), rng.choice(M4 * N4, 3_000_000, replace=False)]))
sc = sp.COO(np.stack([lin // N4, lin % N4]).astype(np.int32), rng.random(lin.size).astype(np.float32), shape=(M4, N4))
sp.sddmm(sc, a4, bt=b4)
p = sc._sddmm_plan[("tiles", K.SDDMM_TILE_THRESHOLD)]
w4 = K.sddmm_panel_width(b4)
restp = K.sddmm_panels(sc.coords, sc.shape, w4, subset=p.rest)
allp = K.sddmm_panels(sc.coords, sc.shape, w4)
t_auto = timeit(lambda: K.sddmm_coo_mfma(p, sc.coords, sc.shape, sc.data, a4, b4, force=True, rest_panels=restp))
t_samp = timeit(lambda: K.sddmm_coo(sc.coords, sc.data, a4, b4, panels=allp))
flops_dense = 2.0 * p.tiles.numel() * 32 * 32 * Kd
print(f"clustered mask ({lin.size} samples, {p.tiles.numel()} dense tiles holding {p.n_dense_samples}; sddmm_tiles_pay says "
      f"{K.sddmm_tiles_pay(p, a4, b4, w4)}): tiles forced + sampled rest {t_auto:.3f} ms "
      f"({flops_dense / t_auto * 1e-9:.1f} TFLOP/s of tile products), sampled kernel alone (panel order) {t_samp:.3f} ms -> {t_samp / t_auto:.2f}x")
