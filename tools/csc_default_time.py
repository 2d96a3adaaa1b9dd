"""The reference's default GCXS compression of a tall matrix is by columns (argmin(shape)): what does the first and the
steady-state `a @ b` cost when A arrives that way?"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_csr_device
import sparse_amd as sp
from sparse_amd import _settings
_settings.NAN_CHECK = False
M, Kd, N = 1_000_000, 10_000, 128
data, idx, ptr = make_csr_device(M, Kd, 0.01, seed=3)
a_csr = sp.GCXS((data, idx, ptr), shape=(M, Kd), compressed_axes=(0,))
torch.cuda.synchronize(); t = time.perf_counter()
a_csc = a_csr.change_compressed_axes((1,))
torch.cuda.synchronize(); print(f"CSR -> CSC (1e8 nnz): {(time.perf_counter()-t)*1e3:.2f} ms")
t = time.perf_counter(); a_csc2 = a_csr.change_compressed_axes((1,)); torch.cuda.synchronize(); print(f"CSR -> CSC again (warm allocator): {(time.perf_counter()-t)*1e3:.2f} ms")
del a_csc2
b = torch.rand((Kd, N), device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = a_csc @ b
    torch.cuda.synchronize(); print(f"a_csc @ b call {i}: {(time.perf_counter()-t)*1e3:.2f} ms")
ref = a_csr @ b
print("same result as the CSR operand:", torch.equal(r, ref))
