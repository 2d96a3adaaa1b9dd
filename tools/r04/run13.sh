cd /root/repo
timeout 300 python -m pytest tests/test_spgemm_bitmap_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 400 python tools/r04/spgemm_f64.py 2>&1 | tail -3
