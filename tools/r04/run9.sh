cd /root/repo
mkdir -p gpurun_out/r04q
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04q/bench.log 2>&1
tail -1 gpurun_out/r04q/bench.log > gpurun_out/r04q/line.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04q/line.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'shard8', d['roofline'].get('shard_ms_at_world8'))
for k, v in d.get('paths', {}).items():
    if isinstance(v, dict) and 'ms' in v: print(f"{k:30s} {v['ms']:8.3f} {v['frac']:.3f}", v.get('accounting_error', ''))
    else: print(k, v)
print('cpu', d.get('cpu_baseline'))
PY
