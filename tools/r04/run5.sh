cd /root/repo
mkdir -p gpurun_out/r04e
timeout 600 python -m pytest tests/test_spmm_tiled_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/f64_time.py 2>&1 | tail -4
timeout 300 python tools/f64_wide.py 2>&1 | tail -6
timeout 900 python bench_paths.py --rows A1_first 2>&1 | tail -8
