import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from sparse_amd import _kernels as K
from util import random_csr, random_dense
def P(*a): print(*a, flush=True)
mode = sys.argv[1]
M, Kd, dens = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
data, idx, ptr = random_csr(M, Kd, dens, 0, np.float32, np.int32)
b = random_dense(Kd, 128, 1, np.float32)
td, ti, tp, tb = (torch.from_numpy(x).cuda() for x in (data, idx, ptr, b))
if mode == "twopass":
    K.TILED_ONE_PASS_INSPECTOR = False
lay = K.csr_tiled_layout(td, ti, tp, M, Kd, force_sort=(mode == "sort")); torch.cuda.synchronize()
P(mode, "inspector ok; group_ends", lay.group_ends)
K.TILED_ONE_PASS_INSPECTOR = False
ref_lay = K.csr_tiled_layout(td, ti, tp, M, Kd); torch.cuda.synchronize()
if lay.group_ends:
    # compare the one-pass stream with the two-pass one list by list
    rg, kb, gpb, epb, slack, _, _ = K.tiled_params(torch.float32)
    nt = -(-Kd // kb)
    bo = lay[1].cpu().numpy().reshape(-1, nt + 1); ro = ref_lay[1].cpu().numpy()
    s1 = lay[0].cpu().numpy().reshape(-1, 16); s2 = ref_lay[0].cpu().numpy().reshape(-1, 16)
    bad = 0
    for g in range(bo.shape[0]):
        for t in range(nt):
            a0, a1 = bo[g, t], bo[g, t + 1]; r0, r1 = ro[g * nt + t], ro[g * nt + t + 1]
            if a1 - a0 != r1 - r0 or not np.array_equal(s1[a0:a1], s2[r0:r1]):
                bad += 1
                if bad < 4: P("list differs", g, t, a0, a1, r0, r1)
    P("lists differing:", bad, "max blk_off", bo.max(), "stream blocks", s1.shape[0])
got = K.dot_csr_ndarray_tiled(lay, (M, 128), Kd, tb); torch.cuda.synchronize(); P("executor ok")
ref = K.dot_csr_ndarray((M, 128), td, ti, tp, tb); P("equal", torch.equal(got, ref))
