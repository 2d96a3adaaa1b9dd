#!/bin/bash
# the CSC inspector after a change: streams byte-identical to the library variant `old`, per-kernel times, the tests
#   gpurun -- 'bash tools/r06/csc_check.sh [rows]'       rows: also bench_paths A2_default with both libraries
mkdir -p gpurun_out/csc
SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_old.so timeout 600 python tools/r06/csc_stream_hash.py > gpurun_out/csc/old.txt 2>&1
timeout 600 python tools/r06/csc_stream_hash.py > gpurun_out/csc/new.txt 2>&1
diff gpurun_out/csc/old.txt gpurun_out/csc/new.txt && echo STREAMS-IDENTICAL
grep -c repeat-same gpurun_out/csc/new.txt
bash tools/r06/pmc.sh cscnew tl_csc "stats" python /root/repo/tools/r05/csc_time.py > /dev/null 2>&1
python tools/r06/kstat.py gpurun_out/pmc_cscnew/kernel_stats.csv tl_csc
timeout 900 python -m pytest tests/test_csc_inspector_gpu.py -x -q -m gpu 2>&1 | tail -2
if [ "$1" == "rows" ]; then
  for v in old ""; do
    if [ -n "$v" ]; then export SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_$v.so; else unset SPAMD_LIB; fi
    echo "== ${v:-new}"; timeout 600 python bench_paths.py --rows A2_default > gpurun_out/csc/rows_${v:-new}.txt 2>&1; python tools/r05/show_rows.py gpurun_out/csc/rows_${v:-new}.txt | grep A2
  done
fi
