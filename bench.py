#!/usr/bin/env python
"""bench.py — BASELINE.json headline metric: GCXS(CSR) x dense SpMM on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): A = CSR 10^6 x 10^4 at 1 % (exactly 10^8 stored
elements, uniform, sorted column indices, int32 indices, explicit compressed_axes=(0,)),
B = dense 10^4 x 128 fp32, C = A @ B dense 10^6 x 128 fp32.  Inputs are generated on the
device (synthetic) and are resident in HBM before the timed region.  One "step" = one SpMM
through the product path (`sparse_amd.matmul`).  The product path caches a K-tiled block stream of A
on the array at its first product (inspector/executor: C ABI `spamd_spmm_tiled*`,
csrc/spmm_tiled.hip); the bench builds it before the warm-up and reports the one-time cost as
`config.preprocess_ms` (`preprocess_warm_ms` with a warm allocator) and the rate of the cache-less kernel
(`spamd_spmm_csr`) as
`config.first_call_gflops` — `value` is the steady state of repeated products with the same A.
`--no-tiled` benches the cache-less kernel only.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling — every rank owns a
10^6-row block of a (N*10^6) x 10^4 matrix (row-block sharding of the compressed axis), B is
row-sharded and each step starts with one RCCL all-gather of B (the path's only exchange
step); C stays row-sharded.  value = N * flops / max-over-ranks time.

The JSON line also carries `roofline` (HBM-bound, algorithmic bytes / measured kernel time;
`traffic` from the committed rocprofv3 PMC pass if present) and, on rank 0 at N=1,
`cpu_baseline`: the oracle's single-thread C restatement of the reference loop timed on
this box's host CPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_csr_device(M, K, density, seed, idx_dtype=torch.int32, dtype=torch.float32, device="cuda"):
    """Uniform-without-replacement sparse matrix with exactly round(M*K*density) stored
    elements, as CSR with sorted column indices (the layout `sparse.random(...,
    format="gcxs", compressed_axes=(0,))` produces), generated on the device."""
    nnz = int(round(M * K * density))
    g = torch.Generator(device=device).manual_seed(seed)
    total = M * K
    x = torch.empty(0, dtype=torch.int64, device=device)
    while x.numel() < nnz:
        need = nnz - x.numel()
        draw = torch.randint(0, total, (int(need * 1.05) + 1024,), generator=g, device=device)
        x = torch.unique(torch.cat([x, draw]))  # sorted
    if x.numel() > nnz:
        drop = torch.randperm(x.numel(), generator=g, device=device)[: x.numel() - nnz]
        keep = torch.ones(x.numel(), dtype=torch.bool, device=device)
        keep[drop] = False
        x = x[keep]
    rows = torch.div(x, K, rounding_mode="floor")
    cols = (x - rows * K).to(idx_dtype)
    counts = torch.bincount(rows, minlength=M)
    indptr = torch.zeros(M + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    data = torch.rand(nnz, generator=g, device=device, dtype=torch.float32).to(dtype)
    return data, cols, indptr.to(idx_dtype)


def algorithmic_bytes(M, K, N, nnz, val_bytes, idx_bytes):
    """nnz*(val+idx) + (M+1)*idx + K*N*val [B once] + M*N*val [C once]."""
    read = nnz * (val_bytes + idx_bytes) + (M + 1) * idx_bytes + K * N * val_bytes
    write = M * N * val_bytes
    return read, write


def cpu_baseline(data, idx, ptr, b, M, N, gpu_out):
    """Oracle (single-thread C restatement of `_dot_csr_ndarray`) on the host CPU, same inputs."""
    from oracle import build as obuild, oracle

    obuild.build()
    h = [t.cpu().numpy() for t in (data, idx, ptr, b)]
    t0 = time.perf_counter()
    out = oracle.dot_csr_ndarray((M, N), *h)
    dt = time.perf_counter() - t0
    flops = 2.0 * h[0].shape[0] * N
    got = gpu_out.cpu().numpy()
    denom = np.maximum(np.abs(out), 1e-30)
    rel = float(np.max(np.abs(got - out) / denom))
    return {
        "value": flops / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
        "sample": f"full workload once: M={M} nnz={h[0].shape[0]} N={N} fp32, {dt:.2f} s on 1 of "
                  f"{os.cpu_count()} host cores (oracle/oracle.c, gcc -O3 -ffp-contract=off)",
        "seconds": dt, "gpu_vs_cpu_max_rel_err": rel,
    }


def load_traffic(kernel_substr):
    """HBM-side bytes per launch of the named kernel from the committed rocprofv3 PMC pass
    (profiles/traffic.json: {"kernels": {substr: {"traffic_bytes_per_launch": ...}}}), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t["kernels"][kernel_substr]["traffic_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--cols", type=int, default=10_000)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--idx", choices=["int32", "int64"], default="int32")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--exact", action="store_true", help="bit-exact mul+add instead of FMA")
    ap.add_argument("--no-tiled", action="store_true", help="row-group kernel only (no cached block stream)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                             "(one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=device)

    import sparse_amd
    from sparse_amd import _settings

    _settings.NAN_CHECK = False  # the reference's matmul NaN pass is reported separately
    _settings.EXACT_MULADD = bool(args.exact)
    if args.no_tiled:
        _settings.TILED_SPMM = "never"

    M, K, N = args.rows, args.cols, args.n
    idt = torch.int32 if args.idx == "int32" else torch.int64
    data, idx, ptr = make_csr_device(M, K, args.density, seed=1234 + rank, idx_dtype=idt, device=device)
    nnz = int(data.numel())
    a = sparse_amd.GCXS((data, idx, ptr), shape=(M, K), compressed_axes=(0,))
    g = torch.Generator(device=device).manual_seed(99)
    b_full = torch.rand((K, N), generator=g, device=device, dtype=torch.float32)
    if world > 1:
        from sparse_amd import _dist

        b_shard = _dist.row_shard(b_full, rank, world).contiguous()

    def step():
        if world > 1:
            b = _dist.all_gather_rows(b_shard, K)
        else:
            b = b_full
        return sparse_amd.matmul(a, b)

    def dev_time(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # cache-less kernel (what a first product with this A costs), then the one-time inspector
    from sparse_amd import _dot, _kernels

    first_call_ms = preprocess_ms = preprocess_warm_ms = None
    tiled = False
    if _dot._tiled_eligible(a.data, b_full, (M, N), K):
        rowgroup = lambda: _kernels.dot_csr_ndarray((M, N), data, idx, ptr, b_full, exact=False)
        rowgroup()
        first_call_ms = dev_time(rowgroup, 5)
        t_pre = time.perf_counter()
        tiled = _dot.prepare_spmm(a)
        torch.cuda.synchronize()
        preprocess_ms = (time.perf_counter() - t_pre) * 1e3        # first build: includes ~1 GB of fresh allocations
        t_pre = time.perf_counter()
        _kernels.csr_tiled_layout(data, idx, ptr, M, K)
        torch.cuda.synchronize()
        preprocess_warm_ms = (time.perf_counter() - t_pre) * 1e3   # the same build with a warm allocator

    # the NaN pass of the reference's `matmul` (_common.py:245-246) is switched off in the timed region; its cost per
    # product (values of A: memoised on the array after the first product; B: scanned every time) is reported
    _settings.NAN_CHECK = True
    _dot.check_class_nan(a)
    nan_check_ms = dev_time(lambda: (_dot.check_class_nan(a), _dot.check_class_nan(b_full)), 5)
    _settings.NAN_CHECK = False

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    tmax = torch.tensor([wall], device=device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    ms_per_step = wall / args.steps * 1e3

    if rank == 0:
        flops = 2.0 * nnz * N
        rd, wr = algorithmic_bytes(M, K, N, nnz, 4, 4 if idt == torch.int32 else 8)
        kernel_ms = dev_ms / args.steps  # HIP events on the launch stream, kernel(s) only at N=1
        ach = (rd + wr) / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "GCXS x dense SpMM throughput (GFLOP/s)",
            "value": world * flops / (ms_per_step * 1e-3) / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"GCXS(CSR, compressed_axes=(0,)) {M}x{K} @ {args.density:g} "
                            f"({nnz} nnz/GPU, {args.idx} indices) x dense {K}x{N} fp32 -> dense {M}x{N}",
                "per_gpu_rows": M, "nnz_per_gpu": nnz, "idx_dtype": args.idx,
                "parallelism": f"row-block x{world}" + (" + all-gather(B)" if world > 1 else ""),
                "mul_add": "separate (bit-exact)" if args.exact else "fma",
                "kernel": "spmm_tiled (cached block stream)" if tiled else "spmm_csr_rowgroup",
                "preprocess_ms": preprocess_ms, "preprocess_warm_ms": preprocess_warm_ms,
                "first_call_ms": first_call_ms, "nan_check_ms_per_product": nan_check_ms,
                "first_call_gflops": flops / (first_call_ms * 1e-3) / 1e9 if first_call_ms else None,
            },
            "roofline": {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": load_traffic("spmm_tiled" if tiled else "spmm_csr"),
                "algorithmic_bytes": rd + wr, "algorithmic_read_bytes": rd,
                "read_only_frac": rd / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel_ms": kernel_ms, "gflops_per_gpu": flops / (kernel_ms * 1e-3) / 1e9,
            },
        }
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline(data, idx, ptr, b_full, M, N, out)
            except Exception as e:  # the bench line must still be printed
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
