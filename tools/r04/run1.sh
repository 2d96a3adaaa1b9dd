# round 4, GPU call 1: new tests + SpGEMM A/B
cd /root/repo
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_spgemm_bitmap_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a/test_bitmap.log
timeout 600 python tools/r04/spgemm_ab.py 3 > gpurun_out/r04a/spgemm_ab.log 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04a/test_round4.log
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04a/stats -o s -- python /root/repo/tools/r04/spgemm_ab.py 2 > /root/repo/gpurun_out/r04a/stats.log 2>&1 )
cat gpurun_out/r04a/test_bitmap.log gpurun_out/r04a/spgemm_ab.log gpurun_out/r04a/test_round4.log
