# round-5 call 2: new tests (plan path, GCXS same-layout ufuncs, non-canonical B), the whole GPU suite, small workloads after the
# host-side changes, the power-law row
cd /root/repo
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_spgemm_bitmap_gpu.py -x -q -m gpu > gpurun_out/r05/t_new.txt 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r05/t_new.txt
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/t_all.txt 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r05/t_all.txt
timeout 600 python bench_small.py --profile --out gpurun_out/r05/small_2.json > gpurun_out/r05/small_2.txt 2>&1; tail -45 gpurun_out/r05/small_2.txt
timeout 600 python bench_paths.py --rows A1_powerlaw > gpurun_out/r05/powerlaw.txt 2>&1; tail -3 gpurun_out/r05/powerlaw.txt
