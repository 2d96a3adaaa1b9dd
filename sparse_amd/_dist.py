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
    if world <= 1:
        return [0, M]
    # (set-up step, once per matrix: the cut targets are formed on the pointers' own device from nnz = indptr[-1], so the
    # host reads back the world - 1 cuts only — one synchronisation, not one for nnz and one for the cuts)
    ip = indptr.to(torch.int64)
    parts = torch.arange(1, world, dtype=torch.int64, device=ip.device)
    targets = torch.div(parts * ip[-1], world, rounding_mode="floor")
    cuts = torch.searchsorted(ip, targets, right=False)
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


def _start_gather_rows(b_shard, n_rows, group=None):
    """Launch the all-gather of a row-sharded dense operand; returns `finish()`, which waits for the collective (on the
    current stream) and returns the full (n_rows x N) matrix.  Between the two the caller may queue work that does not
    need B (the NaN scan and the inspector of the local block of A run while the shards cross xGMI)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return lambda: b_shard
    cols = b_shard.shape[1:]
    out = torch.empty((n_rows, *cols), dtype=b_shard.dtype, device=b_shard.device)
    if n_rows % world == 0:
        work = dist.all_gather_into_tensor(out, b_shard.contiguous(), group=group, async_op=True)

        def finish_even():
            work.wait()
            return out

        return finish_even
    # ragged shards: pad every shard to the largest one (collectives need equal sizes), gather
    # once, then compact the valid rows into `out`
    maxrows = -(-n_rows // world)
    padded = torch.zeros((maxrows, *cols), dtype=b_shard.dtype, device=b_shard.device)
    padded[: b_shard.shape[0]] = b_shard
    gathered = torch.empty((world * maxrows, *cols), dtype=b_shard.dtype, device=b_shard.device)
    work = dist.all_gather_into_tensor(gathered, padded, group=group, async_op=True)

    def finish_ragged():
        work.wait()
        for r in range(world):
            lo, hi = row_bounds(n_rows, r, world)
            out[lo:hi] = gathered[r * maxrows: r * maxrows + (hi - lo)]
        return out

    return finish_ragged


def all_gather_rows(b_shard, n_rows, group=None):
    """All-gather a row-sharded dense operand into the full (n_rows x N) matrix on every rank.

    Uses one `all_gather_into_tensor` when the shards are equal-sized (the RCCL fast path),
    otherwise one padded gather followed by a local compaction."""
    return _start_gather_rows(b_shard, n_rows, group)()


# OPT-IN memo of the gathered operand (`memo=True`): a static B (inference-style loops) crosses xGMI once, not once per
# product.  Off by default (round-3 advice): the key - pointer, torch version counter, shape, dtype, group - cannot see a
# write that does not bump the version counter (a raw-pointer kernel writing an `out=` buffer, a DLPack / NumPy view), and a
# key that changes on some ranks only sends those ranks into the collective alone.  A caller that opts in owns that
# contract (all ranks update B together, through torch) and calls `invalidate_gather_memo()` after any other write.
_GATHER_MEMO = {}
GATHER_MEMO_ENTRIES = 2


def invalidate_gather_memo():
    """Forget every memoised gathered operand (and release the shards and full matrices it kept alive)."""
    _GATHER_MEMO.clear()


def _group_key(group):
    """A stable identity of a process group: its name when torch gives one, else the global ranks it spans (never `id()`:
    a collected group's id can be reused by a new one)."""
    import torch.distributed as dist

    if group is None:
        return "WORLD"
    name = getattr(group, "group_name", None)
    if name:
        return str(name)
    return tuple(dist.get_process_group_ranks(group))


def _gather_key(t, n_rows, group):
    return (t.data_ptr(), int(t._version), tuple(t.shape), t.dtype, int(n_rows), _group_key(group))


def gathered_rows(b_shard, n_rows, group=None, before_wait=None, memo=False):
    """`all_gather_rows`; `before_wait()` is called after the collective is launched and before it is waited for (also when
    the opt-in memo hits: the caller's preparation work is wanted either way)."""
    key = _gather_key(b_shard, n_rows, group) if memo else None
    hit = _GATHER_MEMO.get(key) if memo else None
    if hit is not None:
        if before_wait is not None:
            before_wait()
        return hit[1]
    finish = _start_gather_rows(b_shard, n_rows, group)
    if before_wait is not None:
        before_wait()
    full = finish()
    if not memo:
        return full
    _GATHER_MEMO[key] = (b_shard, full)       # (the shard is kept alive: a recycled pointer must not alias the key)
    while len(_GATHER_MEMO) > GATHER_MEMO_ENTRIES:
        _GATHER_MEMO.pop(next(iter(_GATHER_MEMO)))
    return full


# Gathers started ahead of their product (`sharded_spmm(..., prefetch=next_shard)`): key of the shard -> (shard, finish).
_PREFETCHED = {}
PREFETCH_ENTRIES = 2


def drop_prefetched():
    """Wait for and forget every gather started ahead (a caller that abandons a loop of prefetched products calls this: a
    collective that was launched must be completed on every rank)."""
    while _PREFETCHED:
        _, (_, finish) = _PREFETCHED.popitem()
        finish()


def sharded_spmm(a_local, b_shard, n_rows_b, group=None, memo=False, prefetch=None):
    """Row-block-sharded C_local = A_local @ all_gather(B): the multi-GPU form of A1/A3.
    `a_local` is this rank's GCXS/COO row block, `b_shard` its slice of B's rows.  While the shards of B are in flight
    the local block is prepared: its NaN scan (`matmul`'s warning, memoised per buffer) and, for an eligible operand,
    its block stream (`prepare_operand`).  `memo=True` opts into the gathered-operand memo (see `_GATHER_MEMO`).

    `prefetch` = the shard of the NEXT product's dense operand (the same tensor when B does not change): its all-gather is
    launched here, BEFORE this product's kernels are queued - RCCL runs it on its own stream behind what the current stream
    held at that moment, i.e. next to this step's executor - and the next call, which finds it by the shard's key, only
    waits for it.  The step's time is then max(product, gather) instead of their sum (at 8 GPUs of config 2 the product is
    ~0.12 ms and a 5 MB all-gather a few tens of microseconds: the difference between 6x and 7x).  Every rank must pass
    `prefetch` in the same calls (the collectives are matched by order), and a loop that ends early calls
    `drop_prefetched()`.  The gathered matrices are distinct buffers: the one a running product reads is never written."""
    from . import _dot

    pending = _PREFETCHED.pop(_gather_key(b_shard, n_rows_b, group), None) if _PREFETCHED else None
    if pending is not None:
        _dot.prepare_operand(a_local, b_shard)
        b = pending[1]()
    else:
        b = gathered_rows(b_shard, n_rows_b, group, before_wait=lambda: _dot.prepare_operand(a_local, b_shard), memo=memo)
    if prefetch is not None:
        _PREFETCHED[_gather_key(prefetch, n_rows_b, group)] = (prefetch, _start_gather_rows(prefetch, n_rows_b, group))
        while len(_PREFETCHED) > PREFETCH_ENTRIES:     # (never silently dropped: an unmatched collective would hang the other ranks)
            _PREFETCHED.pop(next(iter(_PREFETCHED)))[1]()
    return _dot.matmul(a_local, b)


def all_gather_ragged(t, group=None, sizes=None):
    """All-gather 1-D (or [k, n]) tensors whose last dimension differs per rank: one size
    exchange (skipped when the caller already knows `sizes`), one padded `all_gather_into_tensor`, local compaction.
    Returns (cat, sizes)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if sizes is None:
        n = torch.tensor([t.shape[-1]], dtype=torch.int64, device=t.device)
        got = torch.empty(world, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(got, n, group=group)
        sizes = [int(s) for s in got.tolist()]
    if world == 1:
        return t, sizes
    mx = max(max(sizes), 1)
    lead = tuple(t.shape[:-1])
    padded = torch.zeros((*lead, mx), dtype=t.dtype, device=t.device)
    padded[..., : t.shape[-1]] = t
    gathered = torch.empty((world, *lead, mx), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(gathered.view(-1), padded.reshape(-1), group=group)
    return torch.cat([gathered[r][..., : sizes[r]] for r in range(world)], dim=-1), sizes


def _offset_i64(t, value):
    """t + value for an int64 index array: the library's elementwise kernel for device arrays; host arrays (the gloo
    tests of this module's sharding logic run on CPU tensors) are offset by torch."""
    if value == 0:
        return t
    if t.is_cuda:
        from ._umath import binary_arrays

        return binary_arrays("add", t.contiguous(), torch.tensor([value], dtype=torch.int64, device=t.device), b_scalar=True)
    return t + value


def _pad16(n):
    return (n + 15) // 16 * 16


def all_gather_csr(data, indices, indptr, group=None):
    """All-gather row-block shards of a CSR matrix into the whole matrix on every rank — the
    exchange step of row-sharded SpGEMM (B's triplet, 0.8 GB for config 5).
    ONE size exchange ((stored elements, rows) per rank) and ONE collective: every rank packs its values, column indices
    and row-pointer heads (relative to its own first pointer; a shard may be a row-slice view of a larger CSR: only its own
    elements [indptr[0], indptr[-1]) are sent) into one
    byte buffer, 16-byte aligned sections, padded to the largest rank's; the receiver slices the three sections per rank and
    shifts rank r's heads by the stored elements of the ranks before it - the gathered pointers are the whole matrix's
    without a scan over the rows."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = data.device
    rows = int(indptr.numel()) - 1
    if int(indices.numel()) != int(data.numel()):
        raise ValueError("data and indices differ in length")
    # a shard may be a row-slice VIEW of a larger CSR with pointers not rebased: data / indices are then either the shard's
    # own elements or the PARENT's arrays, of which [indptr[0], indptr[-1]) are the shard's (round-4 advice: the parent's
    # arrays were sent from element 0)
    first, last = (int(v) for v in torch.stack([indptr[0], indptr[-1]]).tolist()) if rows >= 0 and indptr.numel() else (0, 0)
    nnz = last - first
    if nnz == int(data.numel()):
        pass                        # data / indices ARE the shard's elements (pointers rebased or not: heads are rebased below)
    elif 0 <= first <= last <= int(data.numel()):
        data, indices = data[first:last], indices[first:last]     # the parent's arrays: the shard's own piece of them
    else:
        raise ValueError("indptr does not describe data / indices (nor a slice of them)")
    got = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(got, torch.tensor([nnz, rows], dtype=torch.int64, device=dev), group=group)
    got = got.tolist()
    nnzs, nrows = [int(v) for v in got[0::2]], [int(v) for v in got[1::2]]
    total = sum(nnzs)
    heads = indptr[:-1].to(torch.int64) - indptr[:1].to(torch.int64)        # rebased to the shard's own first element
    ds, isz = data.element_size(), indices.element_size()
    sect = lambda n, r: (_pad16(n * ds), _pad16(n * isz), _pad16(r * 8))
    width = max(max(sum(sect(n, r)) for n, r in zip(nnzs, nrows)), 16)
    mine = torch.zeros(width, dtype=torch.uint8, device=dev)
    o_d, o_i, o_h = sect(nnz, rows)
    mine[:nnz * ds] = data.contiguous().view(torch.uint8).reshape(-1)
    mine[o_d:o_d + nnz * isz] = indices.contiguous().view(torch.uint8).reshape(-1)
    mine[o_d + o_i:o_d + o_i + rows * 8] = heads.contiguous().view(torch.uint8).reshape(-1)
    if world == 1:
        packed = mine.reshape(1, width)
    else:
        packed = torch.empty((world, width), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(packed.view(-1), mine, group=group)
    d = torch.empty(total, dtype=data.dtype, device=dev)
    i = torch.empty(total, dtype=indices.dtype, device=dev)
    ip = torch.empty(sum(nrows) + 1, dtype=torch.int64, device=dev)
    e0 = r0 = 0
    for r in range(world):
        n, m = nnzs[r], nrows[r]
        s_d, s_i, _ = sect(n, m)
        row = packed[r]
        d[e0:e0 + n] = row[:n * ds].view(data.dtype)
        i[e0:e0 + n] = row[s_d:s_d + n * isz].view(indices.dtype)
        ip[r0:r0 + m] = _offset_i64(row[s_d + s_i:s_d + s_i + m * 8].view(torch.int64), e0)
        e0, r0 = e0 + n, r0 + m
    ip[-1] = total
    wide = indptr.dtype == torch.int64 or total >= 2 ** 31
    return d, i, ip if wide else ip.to(indptr.dtype)


def sharded_spgemm(a_local, b_shard, group=None):
    """C_local = A_local @ all_gather(B): A and B both arrive as row blocks (GCXS,
    compressed_axes=(0,)); C stays row-sharded."""
    from . import _gcxs

    if a_local.compressed_axes != (0,) or b_shard.compressed_axes != (0,):
        raise ValueError("row-block sharding needs compressed_axes=(0,) operands")
    d, i, ip = all_gather_csr(b_shard.data, b_shard.indices, b_shard.indptr, group)
    b_full = _gcxs.GCXS((d, i, ip), shape=(int(ip.numel()) - 1, b_shard.shape[1]), compressed_axes=(0,))
    return a_local @ b_full


def sharded_sddmm(s_local, a_local, bt_shard, n_cols, group=None):
    """out_local = sddmm(S_local, A_local, all_gather(Bt)): mask rows and A rows co-sharded,
    Bt (N x K) row-sharded and gathered once."""
    from . import _api

    bt = gathered_rows(bt_shard, n_cols, group)
    return _api.sddmm(s_local, a_local, bt=bt)
