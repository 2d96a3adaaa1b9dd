cd /root/repo
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r05/t_all.txt 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r05/t_all.txt
timeout 600 python bench_paths.py --rows A1_powerlaw > gpurun_out/r05/powerlaw.txt 2>&1; python tools/r05/show_rows.py gpurun_out/r05/powerlaw.txt
