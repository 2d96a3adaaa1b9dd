cd /root/repo
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_spgemm_bitmap_gpu.py tests/test_matrix_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { lib=$1; shift; if [ -n "$lib" ]; then export SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_$lib.so; else unset SPAMD_LIB; fi; echo "== ${lib:-default} $@"; python tools/r06/stream_time.py "$@" 2>&1 | grep -v amdgpu.ids | sed 's/torch.float/f/; s/rowvec [^|]*| //'; }
run "" 0
run epl4 0
run "" 0
run epl4 0
