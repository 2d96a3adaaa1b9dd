cd /root/repo
timeout 600 python -m pytest tests/test_spmm_tiled_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/f64_time.py 2>&1 | tail -2
timeout 300 python tools/f64_wide.py 2>&1 | tail -3
timeout 300 python tools/r04/coo_first.py 2>&1 | tail -9
