#!/bin/bash
# ONE parameterised script for everything that runs on the GPU box through gpurun (round-4 verdict: rounds 1-4 kept a
# two-to-thirteen-line shell script per call).  From the repository root:
#     gpurun --timeout 1800 -- 'bash tools/gpu_job.sh <job> [args] [-- <job> [args] ...]'
# jobs (several may be chained with `--`; every job's output also lands under gpurun_out/<tag>/, tag = $GPU_JOB_TAG or "job"):
#     tests [pytest args]         python -m pytest <args or `tests`> -x -q -m gpu
#     rows <A1,A9,...>            bench_paths.py --rows <list>, one compact line per row (tools/r05/show_rows.py)
#     bench [bench.py args]       bench.py (default --steps 20 --warmup 5): the last (driver-parsed) line and its size
#     small [args]                bench_small.py (the reference's own benchmark sizes)
#     py <script> [args]          any script of tools/ (timing sweeps, probes)
#     variants <script> <v...>    the script once per library variant (sparse_amd/_lib/variants/libsparse_amd_<v>.so; "" = the default)
#     evidence <rNN>              tools/run_profiles.sh <rNN>: bench line, rocprofv3 stats and PMC passes for profiles/
cd "$(dirname "$0")/.." || exit 1
tag=${GPU_JOB_TAG:-job}
mkdir -p gpurun_out/$tag
run_job() {
  job=$1; shift
  case $job in
    tests)
      args=("$@"); [ ${#args[@]} -eq 0 ] && args=(tests)
      timeout 1500 python -m pytest "${args[@]}" -x -q -m gpu > gpurun_out/$tag/tests.txt 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/$tag/tests.txt ;;
    rows)
      timeout 900 python bench_paths.py --rows "$1" > gpurun_out/$tag/rows.txt 2>&1; python tools/r05/show_rows.py gpurun_out/$tag/rows.txt; grep -i "error\|Traceback" -A3 gpurun_out/$tag/rows.txt | head -12 ;;
    bench)
      args=("$@"); [ ${#args[@]} -eq 0 ] && args=(--steps 20 --warmup 5)
      timeout 900 python bench.py "${args[@]}" > gpurun_out/$tag/bench_stdout.txt 2> gpurun_out/$tag/bench_stderr.txt
      tail -1 gpurun_out/$tag/bench_stdout.txt > gpurun_out/$tag/bench_line.json; wc -c gpurun_out/$tag/bench_line.json
      python -c "import json,sys; d=json.load(open('gpurun_out/$tag/bench_line.json')); print({k: d[k] for k in ('value','ms_per_step')}, {k: d['roofline'].get(k) for k in ('frac','kernel_ms','scaling_proxy')}, d.get('cpu_baseline', {}).get('value'))" ;;
    small)
      timeout 900 python bench_small.py --out gpurun_out/$tag/small_workloads.json "$@" 2>&1 | grep -v amdgpu.ids | tail -40 ;;
    py)
      timeout 900 python "$@" 2>&1 | grep -v amdgpu.ids | tail -60 ;;
    variants)
      script=$1; shift
      for v in "$@"; do
        if [ -n "$v" ]; then export SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_$v.so; else unset SPAMD_LIB; fi
        echo "== ${v:-default}"; timeout 600 python $script 2>&1 | grep -v amdgpu.ids | tail -8
      done; unset SPAMD_LIB ;;
    evidence)
      bash tools/run_profiles.sh "$1" ;;
    *) echo "unknown job $job"; return 2 ;;
  esac
}
cur=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run_job "${cur[@]}"; cur=(); else cur+=("$a"); fi
done
[ ${#cur[@]} -gt 0 ] && run_job "${cur[@]}"
