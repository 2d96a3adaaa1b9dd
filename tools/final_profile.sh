set -x
cd /root/repo
timeout 300 python -m pytest tests/test_spmm_tiled_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/final
timeout 300 python bench.py --steps 30 --warmup 3 2>&1 | tail -1 > gpurun_out/final/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/final/stats -o b -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu > /root/repo/gpurun_out/final/stats.log 2>&1 )
bash tools/tools_pmc.sh final2 spmm_tiled fetch write tcc > gpurun_out/final/pmc.json 2>&1
timeout 600 python bench_paths.py > gpurun_out/final/paths.log 2>&1
ls -R gpurun_out/final | head -30
