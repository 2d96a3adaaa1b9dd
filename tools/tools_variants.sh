#!/bin/bash
# run bench for several SpMM kernel variants (tuning hook), one line each
mkdir -p gpurun_out
for v in "$@"; do
  echo -n "$v : "
  SPAMD_SPMM_VARIANT="$v" python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), 'ms  frac', round(d['roofline']['frac'],4), ' GFLOP/s', round(d['value']))"
done
