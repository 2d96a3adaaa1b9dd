import sys, os
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from sparse_amd import _kernels as K
from util import random_csr, random_dense
M, Kd, dens = 5000, 10000, 0.01
data, idx, ptr = random_csr(M, Kd, dens, 0, np.float32, np.int32)
b = random_dense(Kd, 128, 1, np.float32)
td, ti, tp, tb = (torch.from_numpy(x).cuda() for x in (data, idx, ptr, b))
ref = K.dot_csr_ndarray((M, 128), td, ti, tp, tb)
for rep in range(int(sys.argv[2])):
    lay = K.csr_tiled_layout(td, ti, tp, M, Kd)
    got = K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, tb); torch.cuda.synchronize()
    assert torch.equal(got, ref)
    junk = torch.empty(int(1e6 * (rep % 7 + 1)), device="cuda")   # move the allocations around
print("ok", sys.argv[2], "reps", K.__file__, flush=True)
