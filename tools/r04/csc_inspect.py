"""The CSC-native inspector: products from its block stream against the row-group kernel on the CSR arrays (bit for bit), and
its time against CSC -> CSR + the CSR inspector."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K

def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

cases = [(3000, 700, 0.02, 128), (70_001, 1500, 0.01, 128), (560 * 7 + 13, 10_000, 0.003, 128), (200_000, 4000, 0.01, 256),
         (1_000_000, 10_000, 0.01, 128)]
for dt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
    for (M, Kd, dens, N) in cases:
        N = N if dt == torch.float32 else N // 2
        data, idx, ptr = make_csr_device(M, Kd, dens, seed=3, dtype=dt)
        idx, ptr = idx.to(it), ptr.to(it)
        cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)        # CSC arrays
        b = torch.rand((Kd, N), device="cuda", dtype=dt) - 0.5
        lay = K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt)
        got = K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b)
        want = K.dot_csr_ndarray((M, N), data, idx, ptr, b)
        ok = torch.equal(got, want)
        t_csc = t(lambda: K.csc_tiled_layout(cd, ci, cp, M, Kd, dtype=dt))
        t_old = t(lambda: K.csr_tiled_layout(*K.csx_swap_2d(cd, ci, cp, Kd, M), M, Kd, dtype=dt, defer_check=True))
        print(f"{str(dt):14s} {M}x{Kd} @ {dens} ({data.numel()} nnz): identical {ok}; CSC inspector {t_csc:.3f} ms, CSC->CSR + CSR inspector {t_old:.3f} ms", flush=True)
        del lay, got, want, cd, ci, cp
# rows out of order inside a column: reported, lists empty
M, Kd = 5000, 800
data, idx, ptr = make_csr_device(M, Kd, 0.02, seed=5)
cd, ci, cp = K.csx_swap_2d(data, idx, ptr, M, Kd)
ci2 = ci.clone(); a0 = int(cp[3]); ci2[a0], ci2[a0 + 1] = ci[a0 + 1].clone(), ci[a0].clone()
lay = K.csc_tiled_layout(cd, ci2, cp, M, Kd)
try:
    K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, torch.rand((Kd, 128), device="cuda"))
    print("unsorted rows: NOT reported")
except K.UnsortedColumns:
    print("unsorted rows: reported")
