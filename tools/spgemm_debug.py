import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
g = sp.random((n4, n4), density=1e-3, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
K.SPGEMM_ROW_LOCAL = True
cl = g @ g
K.SPGEMM_ROW_LOCAL = False
cg = g @ g
ip = cl.indptr.long().cpu().numpy()
il, ig = cl.indices.long().cpu().numpy(), cg.indices.long().cpu().numpy()
dl, dg = cl.data.cpu().numpy(), cg.data.cpu().numpy()
bad = np.nonzero(il != ig)[0]
print("mismatching index entries:", bad.size, "of", il.size, "; data mismatches:", int((dl != dg).sum()))
if bad.size:
    rows = np.unique(np.searchsorted(ip, bad, side="right") - 1)
    print("rows affected:", rows.size, rows[:10])
    r = rows[0]
    a, b = ip[r], ip[r + 1]
    print("row", r, "nnz", b - a, "sorted local?", bool(np.all(np.diff(il[a:b]) > 0)), "sorted global?", bool(np.all(np.diff(ig[a:b]) > 0)))
    first = bad[bad >= a][0] - a
    print("first diff at", first, il[a + first - 2:a + first + 6], ig[a + first - 2:a + first + 6])
    print("same set?", set(il[a:b]) == set(ig[a:b]), "dups local", (b - a) - len(set(il[a:b])))
    prod = None
