"""CSC -> CSR of config 2's matrix (10^8 stored elements; n_minor = 10^6 rows: 20 key bits), f32/int32 and f64/int64."""
import sys, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K
M, Kd = 1_000_000, 10_000
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=1)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r
for vt, it in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
    d, i, p = data.to(vt), idx.to(it), ptr.to(it)
    ms1, csc = t(lambda: K.csx_swap_2d(d, i, p, M, Kd))       # CSR -> CSC (14 key bits)
    ms2, back = t(lambda: K.csx_swap_2d(*csc, Kd, M))          # CSC -> CSR (20 key bits)
    ok = all(torch.equal(x, y) for x, y in zip(back, (d, i, p)))
    print(f"{vt} {it}: CSR->CSC {ms1:.3f} ms, CSC->CSR {ms2:.3f} ms, round trip identical {ok}", flush=True)
