"""host enqueue cost per step of a rank's share at world size 8 (config 2): matmul alone, and sharded_spmm with a world-size-1 RCCL group"""
import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, sparse_amd as sp
from sparse_amd import _dist, _settings
from bench import make_csr_device
_settings.NAN_WARNING = "deferred"
M, Kd, N = 1_000_000, 10_000, 128
d, i, p = make_csr_device(M, Kd, 0.01, 1234)
b8 = _dist.partition_rows_by_nnz(p, 8)
d8, i8, p8, s0, s1 = _dist.shard_csr(d, i, p, 0, 8, b8)
a = sp.GCXS((d8.contiguous(), i8.contiguous(), p8.contiguous()), shape=(s1 - s0, Kd), compressed_axes=(0,))
b = torch.rand((Kd, N), device="cuda")
for _ in range(20): sp.matmul(a, b)
torch.cuda.synchronize()
for reps in (50, 2000):
    t0 = time.perf_counter()
    for _ in range(reps): sp.matmul(a, b)
    enq = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e6
    sp.flush_warnings()
    print(f"matmul x{reps}: enqueue {enq:.1f} us, wall {wall:.1f} us per step", flush=True)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
out = torch.empty_like(b)
for _ in range(10): dist.all_gather_into_tensor(out, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    dist.all_gather_into_tensor(out, b); sp.matmul(a, out)
enq = (time.perf_counter() - t0) / 500 * 1e6
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 500 * 1e6
print(f"all_gather(world 1) + matmul: enqueue {enq:.1f} us, wall {wall:.1f} us per step", flush=True)
dist.destroy_process_group()
