cd /root/repo
mkdir -p gpurun_out/r05
for v in "kernel:" "passes:" "kernel:rowcache"; do
  export SPAMD_SDDMM_HALVES=${v%%:*}; export SPAMD_SDDMM_1K=${v##*:}
  echo "== halves=$SPAMD_SDDMM_HALVES 1k=$SPAMD_SDDMM_1K"
  timeout 600 python bench_paths.py --rows A9_sddmm,A9 > gpurun_out/r05/a9_$SPAMD_SDDMM_HALVES$SPAMD_SDDMM_1K.txt 2>&1; python tools/r05/show_rows.py gpurun_out/r05/a9_$SPAMD_SDDMM_HALVES$SPAMD_SDDMM_1K.txt | grep "A9_sddmm"
done
unset SPAMD_SDDMM_HALVES SPAMD_SDDMM_1K
timeout 900 python -m pytest tests/test_sddmm_gpu.py tests/test_round5_gpu.py -x -q -m gpu > gpurun_out/r05/t_sddmm.txt 2>&1; echo "sddmm tests rc=$?"; tail -4 gpurun_out/r05/t_sddmm.txt
