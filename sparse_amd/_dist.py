"""Row-block sharding of the compressed axis across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI).  SpMM /
tensordot / SDDMM output rows are independent, so A is split into nnz-balanced row blocks
(binary search on indptr), C stays row-sharded and the only collective on the path is ONE
all-gather of the dense operand B when it arrives sharded.  xGMI is point-to-point (7 links
x ~153 GB/s per GPU): B (5 MB for the headline config) is gathered in a single collective —
never the 512 MB output.
"""
import torch


def partition_rows_by_nnz(indptr, world):
    """Row boundaries r[0..world] (r[0]=0, r[world]=M) such that every block holds ~nnz/world
    stored elements: r[p] = first row whose indptr value >= p*nnz/world.  Pure integer logic on
    the indptr array (host or device tensor); deterministic and identical on every rank."""
    M = int(indptr.numel()) - 1
    if M < 0:
        raise ValueError("indptr must have at least one entry")
    nnz = int(indptr[-1]) if M >= 0 and indptr.numel() else 0
    ip = indptr.to(torch.int64)
    targets = torch.tensor([(p * nnz) // world for p in range(1, world)], dtype=torch.int64, device=ip.device)
    cuts = torch.searchsorted(ip, targets, right=False) if world > 1 else targets
    bounds = [0] + [min(int(c), M) for c in cuts.tolist()] + [M]
    for i in range(1, len(bounds)):  # monotone even with long empty stretches
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def shard_csr(data, indices, indptr, rank, world, bounds=None):
    """The rank's row block as (data, indices, rebased indptr, row0, row1) — views, no copy of
    data/indices."""
    if bounds is None:
        bounds = partition_rows_by_nnz(indptr, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    p0, p1 = int(indptr[r0]), int(indptr[r1])
    return data[p0:p1], indices[p0:p1], indptr[r0:r1 + 1] - indptr[r0], r0, r1


def row_bounds(n_rows, rank, world):
    """Even split of a dense operand's rows: block `rank` is [lo, hi)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def row_shard(b, rank, world):
    lo, hi = row_bounds(b.shape[0], rank, world)
    return b[lo:hi]


def all_gather_rows(b_shard, n_rows, group=None):
    """All-gather a row-sharded dense operand into the full (n_rows x N) matrix on every rank.

    Uses one `all_gather_into_tensor` when the shards are equal-sized (the RCCL fast path),
    otherwise one padded gather followed by a local compaction."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return b_shard
    cols = b_shard.shape[1:]
    out = torch.empty((n_rows, *cols), dtype=b_shard.dtype, device=b_shard.device)
    if n_rows % world == 0:
        dist.all_gather_into_tensor(out, b_shard.contiguous(), group=group)
        return out
    # ragged shards: pad every shard to the largest one (collectives need equal sizes), gather
    # once, then compact the valid rows into `out`
    maxrows = -(-n_rows // world)
    padded = torch.zeros((maxrows, *cols), dtype=b_shard.dtype, device=b_shard.device)
    padded[: b_shard.shape[0]] = b_shard
    gathered = torch.empty((world * maxrows, *cols), dtype=b_shard.dtype, device=b_shard.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    for r in range(world):
        lo, hi = row_bounds(n_rows, r, world)
        out[lo:hi] = gathered[r * maxrows: r * maxrows + (hi - lo)]
    return out


def sharded_spmm(a_local, b_shard, n_rows_b, group=None):
    """Row-block-sharded C_local = A_local @ all_gather(B): the multi-GPU form of A1/A3.
    `a_local` is this rank's GCXS/COO row block, `b_shard` its slice of B's rows."""
    from ._dot import matmul

    b = all_gather_rows(b_shard, n_rows_b, group)
    return matmul(a_local, b)


def all_gather_ragged(t, group=None):
    """All-gather 1-D (or [k, n]) tensors whose last dimension differs per rank: one size
    exchange, one padded `all_gather_into_tensor`, local compaction.  Returns (cat, sizes)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[-1]], dtype=torch.int64, device=t.device)
    sizes = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = [int(s) for s in sizes.tolist()]
    if world == 1:
        return t, sizes
    mx = max(max(sizes), 1)
    lead = tuple(t.shape[:-1])
    padded = torch.zeros((*lead, mx), dtype=t.dtype, device=t.device)
    padded[..., : t.shape[-1]] = t
    gathered = torch.empty((world, *lead, mx), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(gathered.view(-1), padded.reshape(-1), group=group)
    return torch.cat([gathered[r][..., : sizes[r]] for r in range(world)], dim=-1), sizes


def all_gather_csr(data, indices, indptr, group=None):
    """All-gather row-block shards of a CSR matrix into the whole matrix on every rank — the
    exchange step of row-sharded SpGEMM (B's triplet, 0.8 GB for config 5)."""
    d, sizes = all_gather_ragged(data, group)
    i, _ = all_gather_ragged(indices, group)
    counts, rsizes = all_gather_ragged((indptr[1:] - indptr[:-1]).to(torch.int64), group)
    ip = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=data.device)
    ip[1:] = torch.cumsum(counts, 0)
    return d, i, ip.to(indptr.dtype) if indptr.dtype == torch.int64 or int(ip[-1]) < 2 ** 31 else ip


def sharded_spgemm(a_local, b_shard, group=None):
    """C_local = A_local @ all_gather(B): A and B both arrive as row blocks (GCXS,
    compressed_axes=(0,)); C stays row-sharded."""
    from ._gcxs import GCXS

    if a_local.compressed_axes != (0,) or b_shard.compressed_axes != (0,):
        raise ValueError("row-block sharding needs compressed_axes=(0,) operands")
    d, i, ip = all_gather_csr(b_shard.data, b_shard.indices, b_shard.indptr, group)
    b_full = GCXS((d, i, ip), shape=(int(ip.numel()) - 1, b_shard.shape[1]), compressed_axes=(0,))
    return a_local @ b_full


def sharded_sddmm(s_local, a_local, bt_shard, n_cols, group=None):
    """out_local = sddmm(S_local, A_local, all_gather(Bt)): mask rows and A rows co-sharded,
    Bt (N x K) row-sharded and gathered once."""
    from ._api import sddmm

    bt = all_gather_rows(bt_shard, n_cols, group)
    return sddmm(s_local, a_local, bt=bt)
