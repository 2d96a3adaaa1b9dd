"""CSR -> CSC (change_compressed_axes) and COO -> GCXS at 10^7 stored elements, for `rocprofv3 --kernel-trace --stats`."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
g = sp.random((100_000, 10_000), density=0.01, random_state=5, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
which = sys.argv[1] if len(sys.argv) > 1 else "csc"
for _ in range(5):
    if which == "csc":
        K.csx_swap_2d(g.data, g.indices, g.indptr, 100_000, 10_000)
    else:
        c = g.tocoo(); c.asformat("gcxs", compressed_axes=(1,))
torch.cuda.synchronize()
