cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_sddmm_gpu.py tests/test_round5_gpu.py tests/test_round4_gpu.py tests/test_round2_gpu.py -x -q -m gpu > gpurun_out/r05/t_sddmm.txt 2>&1; echo "sddmm tests rc=$?"; tail -8 gpurun_out/r05/t_sddmm.txt
timeout 600 python bench_paths.py --rows A9 > gpurun_out/r05/a9.txt 2>&1; python tools/r05/show_rows.py gpurun_out/r05/a9.txt
