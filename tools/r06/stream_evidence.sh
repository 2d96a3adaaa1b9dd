#!/bin/bash
# counters and per-wave stamps of the stream kernel against the row-vector kernel (profiles/r06_rowvec_pmc.json, r06_stream_waves.txt)
cd /root/repo
for cfg in "n1_stream 1 f32" "n2_stream 2 f32" "n4_stream 4 f32" "n1_f64_stream 1 f64" "n1_rowvec 1 f32 rowvec" "n2_rowvec 2 f32 rowvec" "n4_rowvec 4 f32 rowvec"; do
  set -- $cfg; tag=$1; shift
  bash tools/r06/pmc.sh ev_$tag "spmm_" "stats sq sq3 fetch write" python /root/repo/tools/r06/stream_one.py "$@" > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, os
out = {"what": "rocprofv3 passes (tools/r06/pmc.sh: --kernel-trace --stats, then one --pmc group per run) over tools/r06/stream_one.py: config 2's matrix (10^6 x 10^4 @ 1 %, fp32 / int32) x dense 10^4 x N, six launches; *_stream = spmm_stream.hip (round 6), *_rowvec = the row-vector kernel of spmm_csr.hip it replaces (SPAMD_SPMM_ROWVEC); per-dispatch averages; fabric reads = 2 x FETCH_SIZE x 1024 on gfx950", "rows": {}}
for d in sorted(glob.glob("gpurun_out/pmc_ev_*")):
    tag = os.path.basename(d)[7:]
    row = {}
    s = json.load(open(os.path.join(d, "summary.json")))
    row["counters"] = {k: v["avg_per_dispatch"] for k, v in s.items()}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        if "spmm_stream_kernel" in r["Name"] or "rowvec" in r["Name"]:
            row["kernel"], row["rocprof_avg_ms"], row["calls"] = r["Name"][:90], float(r["AverageNs"]) / 1e6, int(r["Calls"])
    c = row["counters"]
    if "SQ_WAVE_CYCLES" in c:
        row["wave_time"] = {"s_waitcnt": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stall": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]}
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        row["lds_conflict_share_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c:
        row["fabric_read_bytes"] = 2 * 1024 * c["FETCH_SIZE"]
    out["rows"][tag] = row
json.dump(out, open("gpurun_out/r06_rowvec_pmc.json", "w"), indent=1)
for k, v in out["rows"].items():
    print(k, v.get("rocprof_avg_ms"), v.get("wave_time"), v.get("lds_conflict_share_of_lds_cycles"), v.get("fabric_read_bytes"))
PY
SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_ph.so PHASES=1 python tools/r06/stream_prof.py 2>&1 | grep -v amdgpu > gpurun_out/r06_stream_waves.txt
tail -5 gpurun_out/r06_stream_waves.txt
