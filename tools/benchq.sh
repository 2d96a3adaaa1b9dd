cd /root/repo
timeout 150 python -m pytest tests/test_spmm_tiled_gpu.py -x -q 2>&1 | tail -3
timeout 120 python bench.py --steps 20 --warmup 3 > gpurun_out/bench2.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/bench2.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "first", d["config"].get("first_call_ms"), "err", d["cpu_baseline"]["gpu_vs_cpu_max_rel_err"])
else:
    print(open("gpurun_out/bench2.log").read()[-1500:])
PY
