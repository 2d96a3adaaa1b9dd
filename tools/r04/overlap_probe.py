"""Do the inspector of one matrix and the executor of another share the chip when enqueued on two streams?"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
from sparse_amd import _kernels as K

M, Kd, N = 1_000_000, 10000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=0)
b = torch.rand((Kd, N), device="cuda")
out = torch.empty((M, N), device="cuda")
lay = K.csr_tiled_layout(data, idx, ptr, M, Kd)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def wall(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

def ex():
    K.dot_csr_ndarray_tiled(lay, (M, N), Kd, b, out=out)
def ins():
    return K.csr_tiled_layout(data, idx, ptr, M, Kd, defer_check=True)
def seq():
    ex(); ins()
def both():
    with torch.cuda.stream(s1): ex()
    with torch.cuda.stream(s2): l = ins()
    return l
print("executor", wall(ex)); print("inspector", wall(ins)); print("sequential", wall(seq)); print("two streams", wall(both))
# quarter-size pieces, pipelined: inspector(c+1) next to executor(c)
Q = 4
rows = [(M * q // Q // 560 * 560 if q < Q else M) for q in range(Q + 1)]
parts = []
for q in range(Q):
    r0, r1 = rows[q], rows[q + 1]
    p0, p1 = int(ptr[r0]), int(ptr[r1])
    parts.append((data[p0:p1], idx[p0:p1], (ptr[r0:r1 + 1] - ptr[r0]).contiguous(), r1 - r0, r0))
def piecewise(pipelined):
    lays = [None] * Q
    ev = [torch.cuda.Event() for _ in range(Q)]
    for q in range(Q):
        d, i, p, m, r0 = parts[q]
        with torch.cuda.stream(s2 if pipelined else s1):
            lays[q] = K.csr_tiled_layout(d, i, p, m, Kd, defer_check=True); ev[q].record()
        with torch.cuda.stream(s1):
            s1.wait_event(ev[q])
            K.dot_csr_ndarray_tiled(lays[q], (m, N), Kd, b, out=out[r0:r0 + m])
    return lays
print("4 pieces, one stream", wall(lambda: piecewise(False))); print("4 pieces, pipelined", wall(lambda: piecewise(True)))
