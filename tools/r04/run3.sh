cd /root/repo
mkdir -p gpurun_out/r04c
timeout 300 python -m pytest tests/test_spgemm_bitmap_gpu.py -q -m gpu 2>&1 | tail -4
SPAMD_LIB=sparse_amd/_lib/variants/libsparse_amd_bmkprof.so SPAMD_BMK_PROF=1 timeout 300 python tools/r04/spgemm_ab.py 2 > gpurun_out/r04c/prof.log 2>&1
cat gpurun_out/r04c/prof.log
timeout 300 python tools/r04/spgemm_ab.py 3 2>&1 | tail -3
