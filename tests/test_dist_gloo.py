"""world_size-2 `gloo` test of the multi-GPU path's host logic: nnz-balanced
row-block sharding of A + one all-gather of the row-sharded dense operand B.  The local
product of each rank is formed here by the ORACLE (test infrastructure) because this container
has no GPU; what is under test is the sharding arithmetic and the collective plumbing that
`sparse_amd._dist` / bench.py use unchanged with the "nccl" (RCCL) backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import random_csr, random_dense


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, K, N, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from sparse_amd import _dist

    M = 501
    data, idx, ptr = random_csr(M, K, 0.05, 9, np.float64, np.int64, empty_rows=(0, 1, 2), long_row=77)
    b = random_dense(K, N, 10, np.float64)
    tptr = torch.from_numpy(ptr)
    bounds = _dist.partition_rows_by_nnz(tptr, world)
    d, i, p, r0, r1 = _dist.shard_csr(torch.from_numpy(data), torch.from_numpy(idx), tptr, rank, world, bounds)
    b_shard = _dist.row_shard(torch.from_numpy(b), rank, world)
    b_full = _dist.all_gather_rows(b_shard, K)
    assert torch.equal(b_full, torch.from_numpy(b))
    local = oracle.dot_csr_ndarray((r1 - r0, N), d.numpy(), i.numpy(), p.numpy(), b_full.numpy())
    whole = oracle.dot_csr_ndarray((M, N), data, idx, ptr, b)
    assert np.array_equal(local, whole[r0:r1])
    # the output stays row-sharded: row counts add up and cover [0, M)
    cnt = torch.tensor([r1 - r0])
    dist.all_reduce(cnt)
    assert int(cnt) == M
    ret[rank] = (r0, r1)
    dist.destroy_process_group()


@pytest.mark.parametrize("K,N", [(300, 8), (301, 5)])  # even and ragged shards of B
def test_row_block_sharding_with_allgather_world2(K, N):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, N, ret), nprocs=world, join=True)
    spans = sorted(ret.values())
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == 501


def _worker_csr(rank, world, port, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparse_amd import _dist

    data, idx, ptr = random_csr(403, 97, 0.08, 21, np.float64, np.int64, empty_rows=(0, 402), long_row=200)
    td, ti, tp = (torch.from_numpy(x) for x in (data, idx, ptr))
    bounds = _dist.partition_rows_by_nnz(tp, world)
    d, i, p, r0, r1 = _dist.shard_csr(td, ti, tp, rank, world, bounds)
    gd, gi, gp = _dist.all_gather_csr(d, i, p)
    assert torch.equal(gd, td) and torch.equal(gi, ti) and torch.equal(gp, tp)
    cat, sizes = _dist.all_gather_ragged(torch.arange(rank + 3))
    assert sizes == [3, 4] and cat.tolist() == [0, 1, 2, 0, 1, 2, 3]
    ret[rank] = True
    dist.destroy_process_group()


def test_allgather_csr_triplet_world2():
    """The exchange step of row-sharded SpGEMM: gathering the row-block shards of B reproduces B."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_csr, args=(2, port, ret), nprocs=2, join=True)
    assert len(ret) == 2


def _worker_sddmm(rank, world, port, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparse_amd import _dist

    # the sharding of `sharded_sddmm`: mask rows and A rows co-sharded by nnz-balanced row blocks,
    # Bt (n_cols x K) row-sharded and gathered once; the local sampled products are evaluated here in NumPy float64
    # (no GPU in this container) — under test are the row blocks and the collective
    M, Ncols, Kd = 211, 157, 24
    data, idx, ptr = random_csr(M, Ncols, 0.06, 31, np.float64, np.int64, empty_rows=(5, 6), long_row=100)
    a = random_dense(M, Kd, 32, np.float64)
    bt = random_dense(Ncols, Kd, 33, np.float64)
    tp = torch.from_numpy(ptr)
    bounds = _dist.partition_rows_by_nnz(tp, world)
    d, i, p, r0, r1 = _dist.shard_csr(torch.from_numpy(data), torch.from_numpy(idx), tp, rank, world, bounds)
    bt_full = _dist.all_gather_rows(_dist.row_shard(torch.from_numpy(bt), rank, world), Ncols)
    assert torch.equal(bt_full, torch.from_numpy(bt))
    a_local = a[r0:r1]
    rows = np.repeat(np.arange(r1 - r0), np.diff(p.numpy()))
    local = d.numpy() * np.einsum("ik,ik->i", a_local[rows], bt_full.numpy()[i.numpy()])
    rows_g = np.repeat(np.arange(M), np.diff(ptr))
    whole = data * np.einsum("ik,ik->i", a[rows_g], bt[idx])
    assert np.array_equal(local, whole[ptr[r0]:ptr[r1]])
    nnz = torch.tensor([float(d.numel())])
    gathered = [torch.zeros_like(nnz) for _ in range(world)]
    dist.all_gather(gathered, nnz)
    ret[rank] = [int(t.item()) for t in gathered]
    dist.destroy_process_group()


def test_sharded_sddmm_plumbing_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_sddmm, args=(2, port, ret), nprocs=2, join=True)
    per_rank = ret[0]
    assert per_rank == ret[1] and abs(per_rank[0] - per_rank[1]) <= 157 + 1   # nnz-balanced up to one (long) row


def test_partition_rows_by_nnz_strong_scaling_balance():
    """bench.py --scaling strong splits ONE matrix into nnz-balanced row blocks: at 8 blocks of a uniform matrix the
    heaviest block is within one row of the mean."""
    from sparse_amd import _dist

    data, idx, ptr = random_csr(4000, 500, 0.05, 41, np.float32, np.int32)
    tp = torch.from_numpy(ptr)
    for world in (1, 2, 4, 8):
        b = _dist.partition_rows_by_nnz(tp, world)
        assert b[0] == 0 and b[-1] == 4000 and all(x <= y for x, y in zip(b, b[1:]))
        per = [int(ptr[b[r + 1]] - ptr[b[r]]) for r in range(world)]
        assert sum(per) == len(data) and max(per) - len(data) / world <= 500
