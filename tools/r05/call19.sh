cd /root/repo; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_spgemm_bitmap_gpu.py tests/test_matrix_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r05/t_small.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05/t_small.txt
python tools/r05/prof_spsp.py 2>&1 | grep "us per call"
