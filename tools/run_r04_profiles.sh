# Round-4 evidence run (one gpurun call).  Everything lands under gpurun_out/r04/:
#   bench_line.json     the driver's command (python bench.py), full line with `paths`
#   stats_headline/     rocprofv3 --kernel-trace --stats of the driver's command without the CPU leg and without `paths`
#   stats/              the same with `paths`: kernel averages of every other row's kernels
#   pmc_headline.json   FETCH/WRITE/TCC/SQ counters of the headline kernel (one counter group per run, --kernel-trace only)
#   pmc_paths/          FETCH_SIZE / WRITE_SIZE / TCC / SQ passes over bench_paths.py (every other section-8 row)
cd /root/repo
mkdir -p gpurun_out/r04
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r04/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04/stats_headline -o b -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu --no-paths > /root/repo/gpurun_out/r04/stats_headline.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04/stats -o b -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu > /root/repo/gpurun_out/r04/stats.log 2>&1 )
bash tools/tools_pmc.sh r04 spmm_tiled fetch write tcc sq > gpurun_out/r04/pmc_headline.json 2>&1
cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r04/pmc_paths
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/r04/pmc_paths/$name -o p -- python $R/bench_paths.py > $R/gpurun_out/r04/pmc_paths/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python $R/tools/r04_summarise.py $R/gpurun_out/r04 > $R/gpurun_out/r04/summary.txt 2>&1
# only summaries travel back (gpurun merges at most 64 MiB): the raw per-dispatch counter tables stay on the box
find $R/gpurun_out/r04 $R/gpurun_out/pmc_r04 -name '*counter_collection.csv' -delete
find $R/gpurun_out/r04 $R/gpurun_out/pmc_r04 -name '*kernel_trace.csv' -delete
ls $R/gpurun_out/r04
