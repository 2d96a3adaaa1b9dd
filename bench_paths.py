#!/usr/bin/env python
"""bench_paths.py — the other rows of SURVEY.md §8(a) at BASELINE.json's config sizes.

bench.py carries the headline metric (config 2, A1).  This script times the remaining
hot-path rows through the product API with inputs resident in HBM, and reports each against
its own roofline (algorithmic bytes per SURVEY.md §8d / kernel time at 8 TB/s):

    python bench_paths.py [--quick]      -> one JSON line per row, also written to
                                            gpurun_out/paths.json

Rows: A7 elementwise add/mul + A8 sum on config 1; A3 3-D COO tensordot (config 3, f64 and
f32); A9 SDDMM (config 4, bf16 in / fp32 acc); A2 CSC x dense incl. the CSC->CSR
re-compression; A6 COO->GCXS conversion; A4 SpGEMM at a single-GPU size.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM = 8000.0


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        r = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


def line(row, workload, ms, bytes_alg, flops=None, **extra):
    d = {"row": row, "workload": workload, "ms": ms, "algorithmic_bytes": bytes_alg,
         "GBps": bytes_alg / ms / 1e6, "frac_hbm_8TBs": bytes_alg / ms / 1e6 / HBM}
    if flops:
        d["GFLOPs"] = flops / ms / 1e6
    d.update(extra)
    print(json.dumps(d), flush=True)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="1/10 sizes")
    args = ap.parse_args()
    q = 10 if args.quick else 1
    import sparse_amd as sp
    from sparse_amd import _settings

    _settings.NAN_CHECK = False
    out = []
    dev = torch.device("cuda")

    # ---- config 1: COO + COO, (1000,1000,1000), 1e6 nnz each, f64 / int64 -----------------------
    nnz = 1_000_000 // q
    x = sp.random((1000, 1000, 1000), nnz=nnz, random_state=0)
    y = sp.random((1000, 1000, 1000), nnz=nnz, random_state=1)
    def with_coords(f):  # results are built on linear keys; `.coords` splits them on demand
        def g():
            r = f()
            r.coords
            return r
        return g

    for name, f in (("add", lambda: x + y), ("multiply", lambda: x * y), ("add + .coords", with_coords(lambda: x + y))):
        ms, z = timed(f)
        b = 2 * nnz * (3 * 8 + 8) + z.nnz * (3 * 8 + 8)
        out.append(line(f"A7 elementwise {name}", f"COO(1000^3, {nnz} nnz, f64/int64) {name} COO", ms, b,
                        out_nnz=z.nnz))
    z = x + y
    ms, s = timed(lambda: z.sum(axis=2))
    out.append(line("A8 reduce sum(axis=2)", f"COO(1000^3, {z.nnz} nnz) sum over last axis", ms,
                    z.nnz * 16 + s.nnz * 16, groups=s.nnz))
    ms, s = timed(lambda: z.sum(axis=0))
    out.append(line("A8 reduce sum(axis=0)", f"COO(1000^3, {z.nnz} nnz) sum over FIRST axis (needs key sort)", ms,
                    z.nnz * 16 + s.nnz * 16, groups=s.nnz))

    # ---- the same rows at 100x the size (streaming regime rather than launch-bound) -----------------
    if not args.quick:
        nb = 100_000_000
        xb = sp.random((1000, 1000, 1000), nnz=nb, random_state=10)
        yb = sp.random((1000, 1000, 1000), nnz=nb, random_state=11)
        for name, f in (("add", lambda: xb + yb), ("multiply", lambda: xb * yb),
                        ("add + .coords", with_coords(lambda: xb + yb))):
            ms, z = timed(f, reps=3)
            out.append(line(f"A7 elementwise {name} (1e8 nnz)", f"COO(1000^3, {nb} nnz each) {name}", ms,
                            2 * nb * 32 + z.nnz * 32, out_nnz=z.nnz))
        ms, s = timed(lambda: xb.sum(axis=2), reps=3)
        out.append(line("A8 reduce sum(axis=2) (1e8 nnz)", "runs of ~100 elements, two-pass grouped reduce", ms, nb * 16 + s.nnz * 16))
        del xb, yb, z, s

    # ---- config 3: 3-D COO tensordot with dense, axes=1 ------------------------------------------
    side = 512 if not args.quick else 128
    nnz3 = int(side ** 3 * 0.01)
    for dt, it in ((np.float64, "int64"), (np.float32, "int32")):
        c3 = sp.random((side, side, side), nnz=nnz3, random_state=2, dtype=dt, idx_dtype=np.dtype(it))
        d = torch.rand((side, side), device=dev, dtype=torch.float64).to(torch.float64 if dt == np.float64 else torch.float32)
        ms, r = timed(lambda: sp.tensordot(c3, d, axes=1))
        vb, ib = np.dtype(dt).itemsize, np.dtype(it).itemsize
        M, N = side * side, side
        b = nnz3 * (2 * ib + vb) + side * N * vb + M * N * vb
        out.append(line(f"A3 tensordot COO x dense ({np.dtype(dt).name}/{it})",
                        f"COO({side}^3 @1%, {nnz3} nnz) . dense({side},{side}), axes=1 (incl. N-D->2-D reshape)", ms, b,
                        flops=2.0 * nnz3 * N))
        # repeated products with the SAME operand: with `enable_caching()` (reference `COO(cache=True)`, core.py:317-338) the
        # reshaped 2-D operand is memoised, so its tiled block stream is built at the second product and reused
        c3.enable_caching()
        for _ in range(3):
            sp.tensordot(c3, d, axes=1)
        ms, r = timed(lambda: sp.tensordot(c3, d, axes=1))
        out.append(line(f"A3 tensordot, cached operand ({np.dtype(dt).name}/{it})",
                        "same product, operand created with caching: tiled executor from the third call on", ms, b,
                        flops=2.0 * nnz3 * N))
        at = c3.reshape((M, side))
        from sparse_amd import _kernels as K

        ms, r = timed(lambda: K.dot_coo_ndarray(at.coords, at.data, d, (M, N)))
        out.append(line(f"A3 kernel only ({np.dtype(dt).name}/{it})", "rows->indptr + CSR kernel on the reshaped operand",
                        ms, b, flops=2.0 * nnz3 * N))

    # ---- config 4: SDDMM mask 1e5 x 1e5 @ 0.1 %, K = 256, bf16 ------------------------------------
    Ms = 100_000 // (3 if args.quick else 1)
    nnz4 = int(Ms * Ms * 0.001)
    s = sp.random((Ms, Ms), nnz=nnz4, random_state=3, dtype=np.float32, idx_dtype=np.int32)
    for dtn, tdt, esz in (("bf16", torch.bfloat16, 2), ("f32", torch.float32, 4)):
        a = torch.rand((Ms, 256), device=dev).to(tdt)
        bt = torch.rand((Ms, 256), device=dev).to(tdt)
        from sparse_amd import _kernels as K

        ms, r = timed(lambda: K.sddmm_coo(s.coords, s.data, a, bt))
        b = nnz4 * (2 * 4 + 4) + 2 * Ms * 256 * esz + nnz4 * 4
        out.append(line(f"A9 SDDMM kernel ({dtn} in, fp32 acc)", f"mask COO({Ms}x{Ms}, {nnz4} nnz) (.) A({Ms}x256) Bt({Ms}x256)",
                        ms, b, flops=2.0 * 256 * nnz4, gather_bytes=nnz4 * 2 * 256 * esz,
                        gather_TBps=nnz4 * 2 * 256 * esz / ms / 1e9))

    # ---- A2 / A6: CSC x dense and format conversion at config-2/10 size ----------------------------
    from bench import make_csr_device

    M2, K2 = 1_000_000 // (10 * q) * 10, 10_000
    M2 = max(M2 // 10, 1000)
    data, idx, ptr = make_csr_device(M2, K2, 0.01, seed=5)
    a_csr = sp.GCXS((data, idx, ptr), shape=(M2, K2), compressed_axes=(0,))
    nn = a_csr.nnz
    ms, a_csc = timed(lambda: a_csr.change_compressed_axes((1,)), reps=3)
    out.append(line("A6 change_compressed_axes CSR->CSC", f"GCXS {M2}x{K2} @1% ({nn} nnz, f32/int32)", ms,
                    2 * nn * 8 + (M2 + K2) * 4))
    bmat = torch.rand((K2, 128), device=dev)
    ms, r = timed(lambda: a_csc @ bmat, reps=3)
    out.append(line("A2 CSC x dense (re-compress + SpMM)", f"GCXS(csc) {M2}x{K2} @1% x dense {K2}x128", ms,
                    nn * 8 + K2 * 128 * 4 + M2 * 128 * 4, flops=2.0 * nn * 128))
    ms, coo = timed(lambda: a_csr.tocoo(), reps=3)
    out.append(line("A6 GCXS->COO", f"{nn} nnz", ms, nn * 8 + nn * 12))
    ms, g = timed(lambda: sp.GCXS.from_coo(coo, compressed_axes=(1,)), reps=3)
    out.append(line("A6 COO->GCXS(ca=1)", f"{nn} nnz (key permute + radix sort + split)", ms, nn * 12 + nn * 8))

    # ---- A1 in float64 (the reference's default dtype) at the config-2 shape ---------------------------------
    if not args.quick:
        d64, i64_, p64 = make_csr_device(1_000_000, 10_000, 0.01, seed=9, dtype=torch.float64)
        a64 = sp.GCXS((d64, i64_, p64), shape=(1_000_000, 10_000), compressed_axes=(0,))
        b64 = torch.rand((10_000, 128), device=dev, dtype=torch.float64)
        a64 @ b64
        ms, r = timed(lambda: a64 @ b64, reps=5)
        nn64 = int(d64.numel())
        out.append(line("A1 GCXS x dense, float64 (tiled kernel)", "GCXS(CSR) 1e6x1e4 @1% (1e8 nnz, f64/int32) x dense 1e4x128",
                        ms, nn64 * 12 + 10_000 * 128 * 8 + 1_000_000 * 128 * 8, flops=2.0 * nn64 * 128))
        del a64, d64, i64_, p64, b64, r
        torch.cuda.empty_cache()

    # ---- A4: SpGEMM at a single-GPU size --------------------------------------------------------------
    torch.cuda.empty_cache()  # the ESC workspace (~38 GB) should not fight the caching allocator
    n4 = 100_000 // q
    g = sp.random((n4, n4), density=1e-4 * (q if args.quick else 1) * 10, random_state=7, dtype=np.float32,
                  idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
    ms, c = timed(lambda: g @ g, reps=3, warm=2)
    prods = float((g.indptr[1:] - g.indptr[:-1]).double()[g.indices.long()].sum())
    out.append(line("A4 SpGEMM G@G (row-local expand-sort-compress in LDS)", f"GCXS {n4}x{n4}, {g.nnz} nnz, {int(prods)} products -> {c.nnz} nnz",
                    ms, g.nnz * 8 + prods * 8 + c.nnz * 8, flops=2.0 * prods))

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "paths.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
