cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/t_all.txt 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r05/t_all.txt
timeout 300 python tools/r05/probe_loop.py 2>&1 | tail -12
timeout 600 python bench_paths.py --rows A1_powerlaw > gpurun_out/r05/powerlaw.txt 2>&1; tail -3 gpurun_out/r05/powerlaw.txt
