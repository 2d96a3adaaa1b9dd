#!/usr/bin/env python
"""bench_small.py — the reference's OWN benchmark workloads (what a drop-in user calls), per call, host overhead included.

    python bench_small.py [--quick] [--profile]      -> gpurun_out/small_workloads.json (+ stdout summary)

Workloads (reference `benchmarks/`, default dtypes: float64 values, int64 coordinates):
  dense   test_benchmark_coo.py:144-176 `test_gcxs_dot_ndarray`: x(m x n, density 0.001) @ t(n x p), m, n, p in {200, 500, 1000},
          x as COO / GCXS compressed_axes (0,) / (1,)
  spsp    test_benchmark_coo.py:9-40   `test_matmul`: x(m x n) @ y(n x p), density 0.01, COO and GCXS
  ewise   test_benchmark_coo.py:48-66  `test_elemwise`: add / mul, side in {100, 500, 1000}, rank 1-4 (side**rank < 2**26), COO / GCXS
  tdot    test_tensordot.py:9-68       `test_tensordot`: dense.coo, coo.coo, coo.dense with m, n, p, q in {10, 50} x {10, 20} x {20, 50} x {10, 50}
  ewise_broadcast  test_benchmark_coo.py:69-94 `test_elemwise_broadcast`: (side, 1, side) add / mul (side, side), density 0.001, COO / GCXS
  ewise_compare    test_elemwise.py:31-33: `operator.gt` of two side x side operands (add / mul of it are `ewise` rank 2)

These sizes hold 40-10^5 stored elements: no kernel of them takes more than a few microseconds, the call is bound by the
host (Python, C-ABI launches, the NaN verdict).  Per workload:
  us_sync   wall-clock of the MEDIAN call with the result COMPLETE on the device before the next call starts (perf_counter
            around call + torch.cuda.synchronize(); the dense operand resident on the device) - a user's latency
  us_pipe   wall-clock per call of a loop that synchronises once at its end - a user's throughput
  us_numpy  (dense workloads) the drop-in form of the reference's benchmark: t is a NumPy array, the result comes back as
            a NumPy array (H2D of t and D2H of the result over PCIe inside every call)
  cpu_us    the oracle's single-core leg on the same operands (the C restatement of the reference's jitted loop; for
            elementwise the NumPy key-union restatement), result compared with the GPU's
`--profile`: cProfile of the 1000 x 1000 x 1000 GCXS-0 @ dense loop + C-ABI calls and torch-visible syncs per call.
"""
import argparse
import gc
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


LAST_SPREAD = {}


def wall(fn, reps, sync_each):
    for _ in range(3):   # (layouts derived on an operand's second or third product, a kernel's first launch: outside the timing)
        fn()
    torch.cuda.synchronize()
    # like `timeit`: no cyclic garbage collection inside the timed loop (a generation-2 pass over this process's objects takes
    # ~35 ms - 700 us per call of a 50-call loop, in whichever workload it happens to fall)
    gc.collect()
    gc.disable()
    global LAST_SPREAD
    try:
        per = []
        t0 = time.perf_counter()
        for _ in range(reps):
            t1 = time.perf_counter()
            r = fn()
            if sync_each:
                torch.cuda.synchronize()
                per.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    if per:
        # us_sync is the MEDIAN call: one call in a few hundred of this process takes 15-80 ms (at no particular call or
        # workload: the same workload alone never shows it - tools/r05/prof_tdot_slow.py), which moves the mean of a 50-call loop by
        # up to 1 ms; the mean and the slowest call are kept beside it
        LAST_SPREAD = {"median_us": round(float(np.median(per)) * 1e6, 1), "max_us": round(max(per) * 1e6, 1),
                       "max_at_call": int(np.argmax(per)), "mean_us": round(dt / reps * 1e6, 1)}
        return LAST_SPREAD["median_us"], r
    return dt / reps * 1e6, r


def cpu_time(fn, budget_s=0.2, max_reps=200):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        r = fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= max_reps:
            return dt / n * 1e6, r


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        return float("inf")
    if got.size == 0:
        return 0.0
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))


def host(t):
    return t if isinstance(t, np.ndarray) else t.cpu().numpy()


def run(quick=False, profile=False, reps=200):
    import sparse_amd as sp
    from sparse_amd import _ffi
    from oracle import oracle

    out = {"_what": __doc__.split("\n")[0], "_reps": reps, "_nan_warning": os.environ.get("SPARSE_AMD_NAN_WARNING", "sync")}
    rng = np.random.default_rng(42)
    sides3 = [(200, 200, 200), (1000, 1000, 1000)] if quick else list(itertools.product([200, 500, 1000], repeat=3))

    # ---- x @ dense --------------------------------------------------------------------------------------------------
    rows = {}
    for (m, n, p) in sides3:
        t_np = rng.random((n, p))
        t_dev = torch.from_numpy(t_np).cuda()
        for fmt in ("coo", "gcxs0", "gcxs1"):
            x = sp.random((m, n), density=0.001, random_state=rng, format="coo")
            if fmt != "coo":
                x = x.asformat("gcxs", compressed_axes=(0,) if fmt == "gcxs0" else (1,))
            us_sync, r = wall(lambda: x @ t_dev, reps, True)
            us_pipe, _ = wall(lambda: x @ t_dev, reps, False)
            us_np, r_np = wall(lambda: x @ t_np, max(reps // 4, 10), False)
            c0 = _ffi.CALLS
            x @ t_dev
            calls = _ffi.CALLS - c0
            # oracle leg: the reference's jitted loop for this format, one core
            if fmt == "coo":
                hc, hd = host(x.coords), host(x.data)
                cpu_us, want = cpu_time(lambda: oracle.dot_coo_ndarray(hc, hd, t_np, (m, p)))
            elif fmt == "gcxs0":
                hd, hi, hp = host(x.data), host(x.indices), host(x.indptr)
                cpu_us, want = cpu_time(lambda: oracle.dot_csr_ndarray((m, p), hd, hi, hp, t_np))
            else:
                hd, hi, hp = host(x.data), host(x.indices), host(x.indptr)
                cpu_us, want = cpu_time(lambda: oracle.dot_csc_ndarray((m, n), (n, p), hd, hi, hp, t_np))
            rows[f"{m}x{n}x{p}_{fmt}"] = {"nnz": int(x.nnz), "us_sync": round(us_sync, 1), "us_pipe": round(us_pipe, 1),
                                         "us_numpy": round(us_np, 1), "c_abi_calls": calls, "cpu_us": round(cpu_us, 1),
                                         "max_rel_err": rel_err(host(r), want), "numpy_result_is_ndarray": isinstance(r_np, np.ndarray)}
    out["dense"] = rows

    # ---- sparse @ sparse --------------------------------------------------------------------------------------------
    rows = {}
    for (m, n, p) in sides3:
        for fmt in ("coo", "gcxs"):
            x = sp.random((m, n), density=0.01, random_state=rng, format=fmt)
            y = sp.random((n, p), density=0.01, random_state=rng, format=fmt)
            us_sync, r = wall(lambda: x @ y, max(reps // 2, 10), True)
            us_pipe, _ = wall(lambda: x @ y, max(reps // 2, 10), False)
            c0 = _ffi.CALLS
            x @ y
            calls = _ffi.CALLS - c0
            xa, ya = x.asformat("gcxs", compressed_axes=(0,)), y.asformat("gcxs", compressed_axes=(0,))
            h = [host(v) for v in (xa.data, ya.data, xa.indices, ya.indices, xa.indptr, ya.indptr)]
            cpu_us, (wd, wi, wp) = cpu_time(lambda: oracle.dot_csr_csr((m, p), *h))
            dense_want = np.zeros((m, p))
            rr = np.repeat(np.arange(m), np.diff(wp))
            np.add.at(dense_want, (rr, wi), wd)
            rows[f"{m}x{n}x{p}_{fmt}"] = {"nnz": [int(x.nnz), int(y.nnz)], "out_nnz": int(r.nnz), "us_sync": round(us_sync, 1),
                                         "us_pipe": round(us_pipe, 1), "c_abi_calls": calls, "cpu_us": round(cpu_us, 1),
                                         "max_rel_err": rel_err(r.todense(), dense_want)}
    out["spsp"] = rows

    # ---- elementwise ------------------------------------------------------------------------------------------------
    rows = {}
    cases = [(s, k) for s, k in itertools.product([100, 500, 1000], [1, 2, 3, 4]) if s ** k < 2 ** 26]
    if quick:
        cases = [(100, 1), (1000, 2)]
    for side, rank in cases:
        for fmt in ("coo", "gcxs"):
            shape = (side,) * rank
            x = sp.random(shape, density=0.01, random_state=rng, format=fmt)
            y = sp.random(shape, density=0.01, random_state=rng, format=fmt)
            xc, yc = x.asformat("coo"), y.asformat("coo")
            hx = (host(xc.linear_loc()), host(xc.data))
            hy = (host(yc.linear_loc()), host(yc.data))
            for name, f, uf in (("add", lambda: x + y, np.add), ("mul", lambda: x * y, np.multiply)):
                us_sync, r = wall(f, reps, True)
                us_pipe, _ = wall(f, reps, False)
                c0 = _ffi.CALLS
                f()
                calls = _ffi.CALLS - c0
                cpu_us, (wk, wv, _, _) = cpu_time(lambda: oracle.elemwise_zero_fill(uf, hx[0], hx[1], hy[0], hy[1]))
                rc = r.asformat("coo")
                same_keys = bool(np.array_equal(host(rc.linear_loc()), wk))
                rows[f"{name}_side{side}_rank{rank}_{fmt}"] = {
                    "nnz": int(x.nnz), "us_sync": round(us_sync, 1), "us_pipe": round(us_pipe, 1), "c_abi_calls": calls,
                    "cpu_us": round(cpu_us, 1), "keys_bit_exact": same_keys,
                    "max_rel_err": rel_err(host(rc.data), wv) if same_keys else None}
    out["ewise"] = rows

    # ---- elementwise with broadcasting, and a comparison (test_benchmark_coo.py:69-94; test_elemwise.py:31-33: add, mul, gt) ----
    rows = {}
    for side in ([100, 1000] if quick else [100, 500, 1000]):
        for fmt in ("coo", "gcxs"):
            x = sp.random((side, 1, side), density=0.001, random_state=rng, format=fmt)
            y = sp.random((side, side), density=0.001, random_state=rng, format=fmt)
            xd, yd = np.asarray(x.todense()), np.asarray(y.todense())
            for name, f, uf in (("add", lambda: x + y, np.add), ("mul", lambda: x * y, np.multiply)):
                us_sync, r = wall(f, reps, True)
                us_pipe, _ = wall(f, reps, False)
                c0 = _ffi.CALLS
                f()
                calls = _ffi.CALLS - c0
                t0 = time.perf_counter()
                want = uf(xd, yd)
                rows[f"{name}_side{side}_{fmt}"] = {"nnz": [int(x.nnz), int(y.nnz)], "out_nnz": int(r.nnz), "us_sync": round(us_sync, 1),
                                                  "us_pipe": round(us_pipe, 1), "c_abi_calls": calls,
                                                  "numpy_dense_us": round((time.perf_counter() - t0) * 1e6, 1),
                                                  "equal_to_numpy_on_the_dense_form": bool(np.array_equal(np.asarray(r.todense()), want))}
    out["ewise_broadcast"] = rows
    rows = {}
    for side in ([1000] if quick else [100, 500, 1000]):
        x = sp.random((side, side), density=0.001, random_state=rng, format="coo") * 10
        y = sp.random((side, side), density=0.001, random_state=rng, format="coo") * 10
        xd, yd = np.asarray(x.todense()), np.asarray(y.todense())
        f = lambda: x > y
        us_sync, r = wall(f, reps, True)
        us_pipe, _ = wall(f, reps, False)
        c0 = _ffi.CALLS
        f()
        rows[f"gt_side{side}_coo"] = {"nnz": [int(x.nnz), int(y.nnz)], "out_nnz": int(r.nnz), "us_sync": round(us_sync, 1),
                                     "us_pipe": round(us_pipe, 1), "c_abi_calls": _ffi.CALLS - c0,
                                     "equal_to_numpy_on_the_dense_form": bool(np.array_equal(np.asarray(r.todense()), xd > yd))}
    out["ewise_compare"] = rows

    # ---- tensordot --------------------------------------------------------------------------------------------------
    rows = {}
    sides4 = list(itertools.product([10, 50], [10, 20], [20, 50], [10, 50]))
    if quick:
        sides4 = [(10, 10, 20, 10), (50, 20, 50, 50)]
    for (m, n, p, q) in sides4:
        t_np = rng.random((m, n))
        t_dev = torch.from_numpy(t_np).cuda()
        cases = {
            "dense.coo": (1, 2, t_dev, lambda: sp.random((m, p, n, q), density=0.01, random_state=rng)),
            "coo.coo": (1, 2, None, lambda: sp.random((m, n, p, q), density=0.01, random_state=rng)),
            "coo.dense": (1, 1, None, lambda: sp.random((m, n, p, q), density=0.01, random_state=rng)),
        }
        for tag, (li, ri, left, mk) in cases.items():
            if tag == "dense.coo":
                lt, rt = t_dev, mk()
            elif tag == "coo.coo":
                lt, rt = sp.random((m, p), density=0.01, random_state=rng), mk()
            else:
                lt, rt = mk(), t_dev
            for rtype, rname in ((np.ndarray, "ndarray"), (sp.COO, "COO")):
                f = lambda: sp.tensordot(lt, rt, axes=([0, li], [0, ri]), return_type=rtype)
                try:
                    us_sync, r = wall(f, max(reps // 2, 10), True)
                    spread = dict(LAST_SPREAD)
                    us_pipe, _ = wall(f, max(reps // 2, 10), False)
                    c0 = _ffi.CALLS
                    f()
                    calls = _ffi.CALLS - c0
                    ld = lt.todense() if hasattr(lt, "todense") else host(lt)
                    rd = rt.todense() if hasattr(rt, "todense") else host(rt)
                    want = np.tensordot(np.asarray(ld), np.asarray(rd), axes=([0, li], [0, ri]))
                    got = r.todense() if hasattr(r, "todense") else host(r)
                    rows[f"{m}-{n}-{p}-{q}_{tag}_{rname}"] = {"us_sync": round(us_sync, 1), "us_sync_mean": spread.get("mean_us"),
                                                             "us_sync_max": spread.get("max_us"), "us_pipe": round(us_pipe, 1), "c_abi_calls": calls,
                                                             "max_abs_err_vs_numpy_dense": float(np.max(np.abs(np.asarray(got) - want))) if want.size else 0.0}
                except Exception as e:  # noqa: BLE001
                    rows[f"{m}-{n}-{p}-{q}_{tag}_{rname}"] = {"error": repr(e)[:200]}
    out["tdot"] = rows

    # ---- summary ----------------------------------------------------------------------------------------------------
    def med(group, key, flt=lambda k: True):
        v = [r[key] for k, r in out[group].items() if key in r and flt(k)]
        return round(float(np.median(v)), 1) if v else None

    out["_summary"] = {
        "dense_1000x1000x1000_gcxs0_us_sync": out["dense"].get("1000x1000x1000_gcxs0", {}).get("us_sync"),
        "dense_1000x1000x1000_gcxs0_us_pipe": out["dense"].get("1000x1000x1000_gcxs0", {}).get("us_pipe"),
        "dense_median_us_sync": med("dense", "us_sync"), "dense_median_us_pipe": med("dense", "us_pipe"),
        "dense_median_us_numpy": med("dense", "us_numpy"), "dense_median_cpu_us": med("dense", "cpu_us"),
        "spsp_median_us_sync": med("spsp", "us_sync"), "spsp_median_cpu_us": med("spsp", "cpu_us"),
        "ewise_median_us_sync": med("ewise", "us_sync"), "ewise_median_us_pipe": med("ewise", "us_pipe"),
        "ewise_median_cpu_us": med("ewise", "cpu_us"),
        "tdot_median_us_sync": med("tdot", "us_sync"),
        "ewise_broadcast_median_us_sync": med("ewise_broadcast", "us_sync"), "ewise_compare_median_us_sync": med("ewise_compare", "us_sync"),
    }

    if profile:
        import cProfile
        import io
        import pstats
        import warnings

        m = n = p = 1000
        x = sp.random((m, n), density=0.001, random_state=1, format="gcxs", compressed_axes=(0,))
        t_dev = torch.rand((n, p), device="cuda", dtype=torch.float64)
        for _ in range(50):
            x @ t_dev
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            x @ t_dev
        enqueue = (time.perf_counter() - t0) / 2000 * 1e6
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / 2000 * 1e6
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            x @ t_dev
        pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
        torch.cuda.set_sync_debug_mode("warn")
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                x @ t_dev
        finally:
            torch.cuda.set_sync_debug_mode("default")
        out["_profile"] = {"case": "GCXS(1000x1000 @ 0.001, compressed_axes=(0,)) @ dense(1000x1000) f64, 2000 calls",
                           "enqueue_us_per_call": round(enqueue, 1), "wall_us_per_call": round(total, 1),
                           "torch_visible_syncs_per_call": sum("synchroniz" in str(v.message).lower() for v in w),
                           "cprofile_top_by_tottime": s.getvalue().splitlines()[:40]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "small_workloads.json"))
    args = ap.parse_args()
    res = run(args.quick, args.profile, args.reps)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["_summary"], indent=1))
    if "_profile" in res:
        print("\n".join(res["_profile"]["cprofile_top_by_tottime"]))
        print({k: v for k, v in res["_profile"].items() if k != "cprofile_top_by_tottime"})


if __name__ == "__main__":
    main()
