cd /root/repo
mkdir -p gpurun_out/r04b
timeout 300 python -m pytest tests/test_spgemm_bitmap_gpu.py -q -m gpu 2>&1 | grep -v "^  File\|Extension modules" | tail -25 > gpurun_out/r04b/test_bitmap.log
cat gpurun_out/r04b/test_bitmap.log
if true; then
timeout 300 python tools/r04/spgemm_ab.py 3 > gpurun_out/r04b/spgemm_ab.log 2>&1
cat gpurun_out/r04b/spgemm_ab.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04b/stats -o s -- python /root/repo/tools/r04/spgemm_ab.py 2 > /root/repo/gpurun_out/r04b/stats.log 2>&1 )
head -12 gpurun_out/r04b/stats/*kernel_stats.csv
fi
