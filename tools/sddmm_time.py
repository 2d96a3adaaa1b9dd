import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
Ms = 100_000; nnz4 = int(Ms * Ms * 0.001)
s = sp.random((Ms, Ms), nnz=nnz4, random_state=3, dtype=np.float32, idx_dtype=np.int32)
for dtn, tdt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    a = torch.rand((Ms, 256), device="cuda").to(tdt); bt = torch.rand((Ms, 256), device="cuda").to(tdt)
    for _ in range(2): r = K.sddmm_coo(s.coords, s.data, a, bt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = K.sddmm_coo(s.coords, s.data, a, bt)
    e1.record(); torch.cuda.synchronize()
    print(dtn, os.environ.get("SPAMD_SDDMM_VARIANT", "default"), f"{e0.elapsed_time(e1)/5:.3f} ms", float(r.double().sum()))
