cd /root/repo
timeout 300 python -m pytest tests/test_spgemm_bitmap_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python tools/r04/spgemm_ab.py 3 2>&1 | grep bitmap
