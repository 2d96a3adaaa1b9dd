cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/spg_stats
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/spg_stats -o s -- python /root/repo/tools/spgemm_time.py > /root/repo/gpurun_out/spg_stats.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/spg_stats/s_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY
tail -3 /root/repo/gpurun_out/spg_stats.log
