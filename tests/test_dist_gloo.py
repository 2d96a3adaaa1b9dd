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


class _Block:
    """Host stand-in for a rank's GCXS row block (this container has no HIP device, so the real container cannot be
    built): carries the CSR triplet and multiplies through the ORACLE.  What runs unmodified is `sparse_amd._dist`."""

    compressed_axes = (0,)
    ndim = 2

    def __init__(self, arg, shape, compressed_axes=(0,), **_):
        self.data, self.indices, self.indptr = arg
        self.shape = tuple(shape)

    def __matmul__(self, other):
        from oracle import oracle

        n = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)   # noqa: E731
        return oracle.dot_csr_csr((self.shape[0], other.shape[1]), n(self.data), n(other.data), n(self.indices),
                                  n(other.indices), n(self.indptr), n(other.indptr))


def _worker_calls_dist(rank, world, port, ret):
    """`sharded_spmm`, `sharded_spgemm` and `sharded_sddmm` THEMSELVES (not a restatement of their bodies) at world size
    2: only the local product of each is replaced by the oracle / NumPy."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from sparse_amd import _api, _dist, _dot, _gcxs

    calls = {"matmul": 0, "prepare": 0, "sddmm": 0}

    def fake_matmul(a, b):
        calls["matmul"] += 1
        return oracle.dot_csr_ndarray((a.shape[0], b.shape[1]), a.data.numpy(), a.indices.numpy(), a.indptr.numpy(), b.numpy())

    def fake_prepare(a, b_like):
        calls["prepare"] += 1

    def fake_sddmm(s, a, bt=None):
        calls["sddmm"] += 1
        rows = np.repeat(np.arange(s.shape[0]), np.diff(s.indptr.numpy()))
        return s.data.numpy() * np.einsum("ik,ik->i", a.numpy()[rows], bt.numpy()[s.indices.numpy()])

    _dot.matmul, _dot.prepare_operand, _api.sddmm, _gcxs.GCXS = fake_matmul, fake_prepare, fake_sddmm, _Block

    # ---- sharded_spmm: ragged B (301 rows over 2 ranks), static B gathered once, changed B gathered again
    M, K, N = 501, 301, 6
    data, idx, ptr = random_csr(M, K, 0.05, 9, np.float64, np.int64, empty_rows=(0, 1, 2), long_row=77)
    b = random_dense(K, N, 10, np.float64)
    tptr = torch.from_numpy(ptr)
    bounds = _dist.partition_rows_by_nnz(tptr, world)
    d, i, p, r0, r1 = _dist.shard_csr(torch.from_numpy(data), torch.from_numpy(idx), tptr, rank, world, bounds)
    a_local = _Block((d, i, p), (r1 - r0, K))
    b_shard = _dist.row_shard(torch.from_numpy(b), rank, world).clone()
    whole = oracle.dot_csr_ndarray((M, N), data, idx, ptr, b)
    gathers = {"n": 0}
    real_start = _dist._start_gather_rows

    def counting_start(*a, **k):
        gathers["n"] += 1
        return real_start(*a, **k)

    _dist._start_gather_rows = counting_start
    for _ in range(2):                               # default: B is gathered at every product (the memo is opt-in)
        assert np.array_equal(_dist.sharded_spmm(a_local, b_shard, K), whole[r0:r1])
    assert gathers["n"] == 2 and calls["matmul"] == 2 and calls["prepare"] == 2, (gathers, calls)
    for _ in range(3):                               # opt-in memo: a static B crosses once
        assert np.array_equal(_dist.sharded_spmm(a_local, b_shard, K, memo=True), whole[r0:r1])
    assert gathers["n"] == 3 and calls["matmul"] == 5 and calls["prepare"] == 5, (gathers, calls)
    b_shard *= 2.0                                   # in-place update on every rank: the version counter changes
    doubled = oracle.dot_csr_ndarray((M, N), data, idx, ptr, 2.0 * b)[r0:r1]
    assert np.array_equal(_dist.sharded_spmm(a_local, b_shard, K, memo=True), doubled)
    assert gathers["n"] == 4
    b_shard.numpy()[...] *= 0.5                      # a write torch's version counter does not see ...
    assert np.array_equal(_dist.sharded_spmm(a_local, b_shard, K, memo=True), doubled)     # ... is served stale by the memo
    _dist.invalidate_gather_memo()                   # the documented remedy
    assert np.array_equal(_dist.sharded_spmm(a_local, b_shard, K, memo=True), whole[r0:r1])
    assert gathers["n"] == 5
    _dist.invalidate_gather_memo()
    # prefetch: step i launches step i + 1's gather before its own product is queued; the next call finds it by the shard's
    # key and starts no gather of its own.  A changing B (a new tensor per step) and an unchanged one (the same tensor).
    g0, m0 = gathers["n"], calls["matmul"]
    shards = [(b_shard * float(k + 1)).clone() for k in range(4)]
    for k in range(4):
        nxt = shards[k + 1] if k + 1 < 4 else None
        want_k = oracle.dot_csr_ndarray((M, N), data, idx, ptr, float(k + 1) * b)[r0:r1]
        assert np.array_equal(_dist.sharded_spmm(a_local, shards[k], K, prefetch=nxt), want_k)
    assert gathers["n"] - g0 == 4 and calls["matmul"] - m0 == 4 and not _dist._PREFETCHED, (gathers, calls)
    g0 = gathers["n"]
    for k in range(3):
        assert np.array_equal(_dist.sharded_spmm(a_local, b_shard, K, prefetch=b_shard), whole[r0:r1])
    assert gathers["n"] - g0 == 4 and len(_dist._PREFETCHED) == 1     # one gather ahead is still pending ...
    _dist.drop_prefetched()                                           # ... and is completed on every rank before the loop is left
    assert not _dist._PREFETCHED
    _dist._start_gather_rows = real_start

    # ---- sharded_spgemm: B's CSR triplet gathered (rebased pointers), local product by the oracle's Gustavson loop
    n2 = 157
    bd, bi, bp = random_csr(n2, n2, 0.07, 21, np.float64, np.int64, empty_rows=(0, n2 - 1), long_row=60)
    tbp = torch.from_numpy(bp)
    bb = _dist.partition_rows_by_nnz(tbp, world)
    sd, si, sp_, s0, s1 = _dist.shard_csr(torch.from_numpy(bd), torch.from_numpy(bi), tbp, rank, world, bb)
    got = _dist.sharded_spgemm(_Block((sd, si, sp_), (s1 - s0, n2)), _Block((sd, si, sp_), (s1 - s0, n2)))
    want = oracle.dot_csr_csr((s1 - s0, n2), sd.numpy(), bd, si.numpy(), bi, sp_.numpy(), bp)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    # a shard whose pointers are NOT rebased (a row-slice view of the whole matrix's pointer array) gathers to the same matrix
    gd, gi, gp = _dist.all_gather_csr(sd, si, tbp[s0:s1 + 1])
    assert np.array_equal(gd.numpy(), bd) and np.array_equal(gi.numpy(), bi) and np.array_equal(gp.numpy(), bp)
    # ... and so does one that hands over the PARENT's data / indices with those pointers (round-4 advice: they were sent from
    # element 0, misaligning every rank's rows but the first's)
    gd, gi, gp = _dist.all_gather_csr(torch.from_numpy(bd), torch.from_numpy(bi), tbp[s0:s1 + 1])
    assert np.array_equal(gd.numpy(), bd) and np.array_equal(gi.numpy(), bi) and np.array_equal(gp.numpy(), bp)

    # ---- sharded_sddmm: mask rows and A rows co-sharded, Bt gathered
    Ms, Nc, Kd = 211, 157, 24
    md, mi, mp_ = random_csr(Ms, Nc, 0.06, 31, np.float64, np.int64, empty_rows=(5, 6), long_row=100)
    a = random_dense(Ms, Kd, 32, np.float64)
    bt = random_dense(Nc, Kd, 33, np.float64)
    tmp = torch.from_numpy(mp_)
    mb = _dist.partition_rows_by_nnz(tmp, world)
    xd, xi, xp, m0, m1 = _dist.shard_csr(torch.from_numpy(md), torch.from_numpy(mi), tmp, rank, world, mb)
    got = _dist.sharded_sddmm(_Block((xd, xi, xp), (m1 - m0, Nc)), torch.from_numpy(a[m0:m1]),
                              _dist.row_shard(torch.from_numpy(bt), rank, world), Nc)
    rows_g = np.repeat(np.arange(Ms), np.diff(mp_))
    assert np.array_equal(got, (md * np.einsum("ik,ik->i", a[rows_g], bt[mi]))[mp_[m0]:mp_[m1]]) and calls["sddmm"] == 1
    ret[rank] = True
    dist.destroy_process_group()


def test_sharded_products_themselves_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_calls_dist, args=(2, port, ret), nprocs=2, join=True)
    assert len(ret) == 2


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
