"""Summarise an evidence run (tools/run_profiles.sh <rNN> -> gpurun_out/<rNN>): per kernel (name prefix) the rocprofv3 average duration and, from the PMC passes over
bench_paths.py, the fabric-side bytes per launch (reads = 2 * FETCH_SIZE * 1024 on gfx950, writes = WRITE_SIZE * 1024:
MI355X_MICROARCH.md, HBM section), the L2 hit rate and the LDS / wave-cycle counters.  -> JSON on stdout."""
import csv, glob, json, os, re, sys
root = sys.argv[1]
def short(n):
    n = re.sub(r"\(.*", "", n)           # drop the argument list
    n = re.sub(r"^void\s+", "", n)
    return n[:140]
stats = {}
for p in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "total_ms": float(r["TotalDurationNs"]) / 1e6}
pmc = {}
seq = {}
for p in glob.glob(os.path.join(root, "pmc_paths", "*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(p)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        d = acc.setdefault(k, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for (kern, ctr), per in acc.items():
        pmc.setdefault(kern, {})[ctr] = sum(per.values()) / len(per)
        pmc[kern]["dispatches"] = len(per)
        seq.setdefault(kern, {})[ctr] = [per[k] for k in sorted(per, key=int)]   # per dispatch, in launch order
out = {}
for kern, c in pmc.items():
    if "spamd" not in kern and "reduce_fill" not in kern and "tl_csc" not in kern:
        continue
    e = dict(c)
    if "FETCH_SIZE" in c: e["fabric_read_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c: e["fabric_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in c and c.get("TCC_REQ_sum"): e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("SQ_LDS_IDX_ACTIVE"): e["lds_conflict_share_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if kern in stats: e["rocprof_avg_ms"] = stats[kern]["avg_ms"]; e["rocprof_calls"] = stats[kern]["calls"]
    out[kern] = e
# ---- rows of bench_paths.py -> the kernels that do their work (one launch each per operation unless stated): the HBM-side
# bytes the row's operation really moved, for bench_paths.py's `pmc_bytes` (profiles/paths_pmc.json).  Only rows whose
# kernels are not shared with rows of another size are listed (rocprofv3 averages a kernel over all its dispatches).
# (kernel-name prefix, launches of it per operation of the row)
ROWS = {
    "A7_1e8_add": [("spamd::mp_partition_kernel", 1), ("spamd::mp_union_kernel<double, double, 2, 4>", 1)],
    "A7_1e8_multiply": [("spamd::mp_partition_kernel", 1), ("spamd::mp_union_kernel<double, double, 2, 4>", 1)],
    "A7_add_config1": [("spamd::mp_union_kernel<double, double, 3, 4>", 1)],
    "A7_multiply_config1": [("spamd::mp_union_kernel<double, double, 3, 4>", 1)],
    "A9_sddmm_bf16": [("spamd::sddmm_panel_kernel<__hip_bfloat16", 1)],
    # round 5: rows of 1 KB = two launches of the panel kernel over 512-byte halves (sddmm_rowcache_kernel on whole rows before)
    "A9_sddmm_f32": [("spamd::sddmm_panel_kernel<float", 2)],
    "A4_spgemm_config5_share": [("spamd::spgemm_bitmap_kernel<float, int, 16, 1024, 512, false>", 1), ("spamd::spgemm_row_products_kernel<int>", 1)],
    "A4_spgemm_config5_share_f64": [("spamd::spgemm_bitmap_kernel<double, long, 8, 512, 256, true>", 1), ("spamd::spgemm_row_products_kernel<long>", 1),
                                    ("spamd::spgemm_bsplit_kernel<long>", 1)],
    "A1_f64": [("spamd::spmm_tiled_kernel<0, 4, double>", 1)],
    "A2_default_gcxs_steady": [("spamd::spmm_tiled_kernel<0, 4, double>", 1)],
    # the reference-default operand's first product: CSC-native inspector (one launch of each kernel) + the float64 executor
    "A2_default_gcxs_first": [("tl_csc_hist_kernel<long>", 1), ("tl_csc_offsets_kernel<5>", 1), ("tl_csc_scan_kernel", 1), ("tl_csc_fill_kernel<long, double>", 1),
                              ("spamd::spmm_tiled_kernel<0, 4, double>", 1)],
}
# rows that share their kernels with another row of a different size: (first, last) share of the kernels' dispatches, in
# launch order (bench_paths.py runs `add` - plain and with coordinates - before `multiply`)
SPLIT = {"A7_1e8_add": (0.0, 0.5), "A7_1e8_multiply": (0.5, 1.0), "A7_add_config1": (0.0, 0.5), "A7_multiply_config1": (0.5, 1.0)}
CACHE_RESIDENT = {"A7_add_config1", "A7_multiply_config1"}   # operands + result < 256 MiB: Infinity-Cache hits are not HBM bytes
rows = {}
for rid, kerns in ROWS.items():
    tot, found = 0.0, []
    for k, per_op in kerns:
        hit = [n for n in out if n.startswith(k)]
        if not hit or "fabric_read_bytes_per_launch" not in out[hit[0]] or "fabric_write_bytes_per_launch" not in out[hit[0]]:
            tot = None
            break
        if rid in SPLIT and hit[0] in seq and "FETCH_SIZE" in seq[hit[0]] and "WRITE_SIZE" in seq[hit[0]]:
            lo, hi = SPLIT[rid]
            f, w = seq[hit[0]]["FETCH_SIZE"], seq[hit[0]]["WRITE_SIZE"]
            f, w = f[int(lo * len(f)):int(hi * len(f))], w[int(lo * len(w)):int(hi * len(w))]
            tot += per_op * (2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w))
        else:
            tot += per_op * (out[hit[0]]["fabric_read_bytes_per_launch"] + out[hit[0]]["fabric_write_bytes_per_launch"])
        found.append(hit[0])
    if tot is not None:
        rows[rid] = {"pmc_bytes": tot, "kernels": found, "cache_resident": rid in CACHE_RESIDENT,
                     "round": os.path.basename(root.rstrip("/"))}
json.dump({"round": os.path.basename(root.rstrip("/")), "rows": rows}, open(os.path.join(root, "paths_pmc.json"), "w"), indent=1)
print(json.dumps({"what": "tools/run_profiles.sh " + os.path.basename(root.rstrip("/")), "kernels": out,
                  "kernel_stats_top": dict(sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:40])}, indent=1))
