cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu -k "balanced" > gpurun_out/r05/t_bal.txt 2>&1; echo "balanced tests rc=$?"; tail -25 gpurun_out/r05/t_bal.txt
timeout 900 python -m pytest tests/test_spmm_tiled_gpu.py tests/test_csc_inspector_gpu.py tests/test_round3_gpu.py -x -q -m gpu > gpurun_out/r05/t_tiled.txt 2>&1; echo "tiled tests rc=$?"; tail -5 gpurun_out/r05/t_tiled.txt
timeout 600 python bench_paths.py --rows A1_powerlaw > gpurun_out/r05/powerlaw.txt 2>&1; python tools/r05/show_rows.py gpurun_out/r05/powerlaw.txt; tail -3 gpurun_out/r05/powerlaw.txt | cut -c1-600
