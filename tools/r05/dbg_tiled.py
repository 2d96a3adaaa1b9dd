import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from sparse_amd import _kernels as K
case = sys.argv[1]
def P(*a): print(*a, flush=True)
if case == "nat":
    from util import random_csr, random_dense
    M, Kd = int(sys.argv[2]), int(sys.argv[3])
    data, idx, ptr = random_csr(M, Kd, float(sys.argv[4]), 0, np.float32, np.int32)
    b = random_dense(Kd, 128, 1, np.float32)
    td, ti, tp, tb = (torch.from_numpy(x).cuda() for x in (data, idx, ptr, b))
    lay = K.csr_tiled_layout(td, ti, tp, M, Kd); torch.cuda.synchronize(); P("inspector ok", lay.group_ends, lay[0].numel(), lay[1].numel())
    got = K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, tb); torch.cuda.synchronize(); P("executor ok")
    ref = K.dot_csr_ndarray((M, 128), td, ti, tp, tb); P("equal", torch.equal(got, ref))
else:
    from bench import make_powerlaw_csr_device
    M, Kd, nnz = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    d, i, p = make_powerlaw_csr_device(M, Kd, nnz, 9)
    b = torch.rand((Kd, 128), device="cuda")
    K.TILED_BALANCE_MIN_NNZ = 0
    lay = K.csr_tiled_layout(d, i, p, M, Kd, defer_check=(case == "bald")); torch.cuda.synchronize(); P("inspector ok", K.TILED_BALANCE_STATS)
    rm = lay.rowmap[: lay.groups * 35].cpu().numpy(); P("rowmap rows", (rm >= 0).sum(), rm.max(), "groups", lay.groups)
    bo = lay[1].cpu().numpy(); P("blk_off max", bo.max(), "blocks", lay[0].numel() // 16)
    got = K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, b); torch.cuda.synchronize(); P("executor ok")
    ref = K.dot_csr_ndarray((M, 128), d, i, p, b); P("equal", torch.equal(got, ref))
