cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu -k "small_spgemm" > gpurun_out/r05/t_small.txt 2>&1; echo "small spgemm tests rc=$?"; tail -12 gpurun_out/r05/t_small.txt
timeout 600 python bench_small.py --quick --out gpurun_out/r05/small_3q.json 2>&1 | grep -v amdgpu | tail -16
