# Round-2 final evidence run (one gpurun call): the bench line with every section-8 row (`paths`), rocprofv3 kernel stats of
# the same command, PMC passes of the headline kernel (one counter group per run), SpGEMM kernel stats, the whole GPU
# test suite.  Lands under gpurun_out/r02f/.
cd /root/repo
mkdir -p gpurun_out/r02f
timeout 400 python bench.py --steps 30 --warmup 3 2>&1 | tail -1 > gpurun_out/r02f/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02f/stats -o b -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu --no-paths > /root/repo/gpurun_out/r02f/stats.log 2>&1 )
bash tools/tools_pmc.sh r02f spmm_tiled fetch write tcc sq > gpurun_out/r02f/pmc.json 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02f/spgemm -o p -- python /root/repo/tools/spgemm_time.py > /root/repo/gpurun_out/r02f/spgemm_time.txt 2>&1 )
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP" | tail -3 > gpurun_out/r02f/gpu_tests.txt
cat gpurun_out/r02f/gpu_tests.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r02f/bench_line.json").read())
print("ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "first", d["config"].get("first_call_ms"), "insp", d["config"].get("inspector_ms"), "err", d["cpu_baseline"]["gpu_vs_cpu_max_rel_err"])
for k, v in d["paths"].items(): print(k, round(v.get("ms", -1), 4), v.get("error", ""))
PY
tail -3 gpurun_out/r02f/spgemm_time.txt
