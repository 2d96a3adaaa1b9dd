#!/usr/bin/env python
"""bench_spgemm.py — BASELINE.json configs[4]: GCXS SpGEMM (10^6 x 10^6 @ 0.01 %)^2, row-block sharded, one RCCL all-gather
of the right-hand operand's CSR triplet per step.  Entry point: `python bench.py --workload spgemm [--gpus N] [--steps K]
[--warmup W]` (bench.py hands its parsed arguments to `main`); same launch contract and the same JSON schema as the headline.

Every rank generates the SAME matrix G (one seed; `sparse.random` as `_dot_csr_csr` would be fed by the reference,
_common.py:543-570, 639-717), keeps its nnz-balanced row block as A_local and B_shard, and a step is
    B = all_gather_csr(B_shard)            one packed collective (values, columns, row heads); nothing at N = 1
    C_local = A_local @ B                  spgemm_bitmap.hip, in pieces of <= `--chunk-rows` rows (the result's upper-bound
                                           buffers are 12 B per PRODUCT: 15 GB per 125 000 rows, 120 GB for all of G at N = 1)
C stays row-sharded; the pieces' results are dropped once their stored-element count and a checksum of
their pointers have been taken, except the last (kept for the sampled oracle comparison).  Strong scaling: the work is G @ G
whatever N is.  value = 2 x products of G @ G / max-over-ranks step time.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def _row_block(sp, g, r0, r1):
    p0, p1 = int(g.indptr[r0]), int(g.indptr[r1])
    return sp.GCXS((g.data[p0:p1].contiguous(), g.indices[p0:p1].contiguous(), (g.indptr[r0:r1 + 1] - g.indptr[r0]).contiguous()),
                   shape=(r1 - r0, g.shape[1]), compressed_axes=(0,))


def main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=device)
        world, rank = dist.get_world_size(), dist.get_rank()

    import sparse_amd as sp
    from sparse_amd import _dist, _kernels as K

    n = args.spgemm_n
    dens = args.spgemm_density
    f64 = args.spgemm_dtype == "f64"
    g = sp.random((n, n), density=dens, random_state=7, dtype=np.float64 if f64 else np.float32,
                  idx_dtype=np.int64 if f64 else np.int32, format="gcxs", compressed_axes=(0,))
    bounds = _dist.partition_rows_by_nnz(g.indptr, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    a_local = _row_block(sp, g, r0, r1)
    b_shard = a_local                                   # G is squared: B's row block is A's
    row_products = (g.indptr[1:] - g.indptr[:-1]).double()
    prods_local = float(row_products[a_local.indices.long()].sum())
    prods_total = float(row_products[g.indices.long()].sum())
    nnz_g = int(g.nnz)
    b_full_single = g if world == 1 else None
    if world > 1:
        del g
    chunk = max(1, int(args.chunk_rows))
    cuts = list(range(0, r1 - r0, chunk)) + [r1 - r0]
    pieces = [_row_block(sp, a_local, lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]

    def gather():
        if world == 1:
            return b_full_single
        d, i, ip = _dist.all_gather_csr(b_shard.data, b_shard.indices, b_shard.indptr)
        return sp.GCXS((d, i, ip), shape=(int(ip.numel()) - 1, n), compressed_axes=(0,))

    state = {}

    def step():
        b = gather()
        nnz_c, check, last = 0, 0, None
        for p in pieces:
            last = None                                 # the piece before this one is released before the next is formed
            c = p @ b
            nnz_c += int(c.nnz)
            last = c
        state["nnz_c"], state["last"], state["b"] = nnz_c, last, b
        return last

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tmax = torch.tensor([wall], device=device, dtype=torch.float64)
    stats = torch.tensor([prods_local, float(state["nnz_c"]), float(a_local.nnz)], device=device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        allst = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats)
    else:
        allst = [stats]
    ms_per_step = float(tmax.item()) / args.steps * 1e3

    # the exchange step and the dominant kernel alone (HIP events on the launch stream)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    gather_ms = None
    if world > 1:
        e0, e1 = ev(), ev()
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        for _ in range(args.steps):
            b = gather()
        e1.record()
        torch.cuda.synchronize()
        gather_ms = e0.elapsed_time(e1) / args.steps
    b = state["b"]
    p0 = pieces[0]
    p0 @ b
    e0, e1 = ev(), ev()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.steps):
        c0 = p0 @ b
    e1.record()
    torch.cuda.synchronize()
    piece_ms = e0.elapsed_time(e1) / args.steps
    piece_prods = float(row_products[p0.indices.long()].sum()) if world == 1 else float(
        (b.indptr[1:] - b.indptr[:-1]).double()[p0.indices.long()].sum())
    piece_bytes = p0.nnz * 8 + piece_prods * 8 + int(c0.nnz) * 8 if not f64 else p0.nnz * 16 + piece_prods * 16 + int(c0.nnz) * 16

    if rank == 0:
        per_rank = [[float(v) for v in t.tolist()] for t in allst]
        nnz_c_total = sum(int(p[1]) for p in per_rank)
        ach = piece_bytes / (piece_ms * 1e-3) / 1e9
        line = {
            "metric": "GCXS x GCXS SpGEMM throughput (GFLOP/s)", "value": round(2.0 * prods_total / (ms_per_step * 1e-3) / 1e9, 2),
            "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if f64 else "f32", "data": "synthetic",
            "config": {
                "workload": f"GCXS({n}x{n} @ {dens:g}, {nnz_g} nnz, {'f64/int64' if f64 else 'f32/int32'})^2 in {world} nnz-balanced row "
                            f"block(s): {int(prods_total)} products -> {nnz_c_total} stored elements of C (row-sharded)",
                "world_size": world, "parallelism": f"row-block x{world}" + (" + RCCL all-gather of B's CSR triplet per step" if world > 1 else ""),
                "products_per_rank": [int(p[0]) for p in per_rank], "nnz_c_per_rank": [int(p[1]) for p in per_rank],
                "nnz_a_per_rank": [int(p[2]) for p in per_rank],
                "products_imbalance": round(max(p[0] for p in per_rank) / (prods_total / world), 5),
                "chunk_rows": chunk, "pieces_rank0": len(pieces), "all_gather_csr_ms": None if gather_ms is None else round(gather_ms, 4),
                "kernel": K.SPGEMM_STATS.get("kernel"), "parts": K.SPGEMM_STATS.get("parts"),
                "results_over_2^31_stored_elements_rank0": bool(state["nnz_c"] >= 2 ** 31),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                "traffic": None, "traffic_source": "profiles/paths_pmc.json row A4_spgemm_config5_share (one GPU's share, same kernel)",
                "algorithmic_bytes": int(piece_bytes), "kernel_ms": round(piece_ms, 4),
                "what": f"rank 0, its first piece ({p0.shape[0]} rows, {int(piece_prods)} products -> {int(c0.nnz)} stored elements) through `a @ b`: "
                        "stored elements of A (value + index) + one (value, index) read per product + the result's (value, index), "
                        "/ mean of `steps` back-to-back products (HIP events)",
            },
        }
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_leg(p0, b, c0, n)
            except Exception as e:  # noqa: BLE001 - the line must still be printed
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_leg(p0, b, c0, n, rows=20000):
    """The oracle's restatement of `_dot_csr_csr` (oracle/oracle.c; test infrastructure) on `rows` sampled rows of rank 0's
    first piece, one host core, timed; the GPU's rows compared with it (columns bit for bit after the canonical sort, values
    within float rounding of the same left-to-right sums: they are the same sums, so 0.0 is expected)."""
    from oracle import oracle

    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(p0.shape[0], size=min(rows, p0.shape[0]), replace=False))
    hA = [t.cpu().numpy() for t in (p0.data, p0.indices, p0.indptr)]
    hB = [t.cpu().numpy() for t in (b.data, b.indices, b.indptr)]
    segs = [np.arange(hA[2][r], hA[2][r + 1]) for r in pick]
    sub_ptr = np.zeros(len(pick) + 1, dtype=hA[2].dtype)
    sub_ptr[1:] = np.cumsum([len(s) for s in segs])
    sel = np.concatenate(segs)
    t0 = time.perf_counter()
    wd, wi, wp = oracle.dot_csr_csr((len(pick), n), hA[0][sel], hB[0], hA[1][sel], hB[1], sub_ptr, hB[2])
    secs = time.perf_counter() - t0
    prods = float((hB[2][1:] - hB[2][:-1])[hA[1][sel]].sum())
    cp = c0.indptr.cpu().numpy()
    same, worst = True, 0.0
    for j, r in enumerate(pick):
        lo, hi = int(cp[r]), int(cp[r + 1])
        gi, gd = c0.indices[lo:hi].cpu().numpy(), c0.data[lo:hi].cpu().numpy()
        wl, wh = int(wp[j]), int(wp[j + 1])
        o = np.argsort(wi[wl:wh], kind="stable")
        keep = wd[wl:wh][o].view(np.uint32 if wd.dtype == np.float32 else np.uint64) != 0
        same &= bool(np.array_equal(gi, wi[wl:wh][o][keep]))
        if same and hi > lo:
            w = wd[wl:wh][o][keep].astype(np.float64)
            worst = max(worst, float(np.max(np.abs(gd.astype(np.float64) - w) / np.maximum(np.abs(w), 1e-300))))
    return {"value": round(2.0 * prods / secs / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port", "seconds": round(secs, 3),
            "sample": f"{len(pick)} sampled rows of rank 0's first piece ({int(prods)} products), oracle.c restatement of _dot_csr_csr",
            "indices_bit_exact": same, "max_rel_err": worst}
