cd /root/repo; mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r05/t_all.txt 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r05/t_all.txt
timeout 900 python bench_small.py --out gpurun_out/r05/small_3.json 2>&1 | grep -v amdgpu | tail -16
