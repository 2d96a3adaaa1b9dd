# Evidence run of a round (one gpurun call): bash tools/run_profiles.sh <rNN>.  Everything lands under gpurun_out/<rNN>/:
#   bench_line.json     the driver's command (python bench.py): its LAST stdout line (what the driver parses); bench_stdout.txt = all of it (the `paths` line first)
#   small_workloads.json  bench_small.py: the reference's own benchmark sizes, per call
#   stats_headline/     rocprofv3 --kernel-trace --stats of the driver's command without the CPU leg and without `paths`
#   stats/              the same with `paths`: kernel averages of every other row's kernels
#   pmc_headline.json   FETCH/WRITE/TCC/SQ counters of the headline kernel (one counter group per run, --kernel-trace only)
#   pmc_paths/          FETCH_SIZE / WRITE_SIZE / TCC / SQ passes over bench_paths.py (every other section-8 row)
cd /root/repo
R5=${1:-r05}
mkdir -p gpurun_out/$R5
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/$R5/bench_stdout.txt 2> gpurun_out/$R5/bench_stderr.txt
tail -1 gpurun_out/$R5/bench_stdout.txt > gpurun_out/$R5/bench_line.json
timeout 900 python bench_small.py --profile --out gpurun_out/$R5/small_workloads.json > gpurun_out/$R5/small_workloads.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$R5/stats_headline -o b -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu --no-paths > /root/repo/gpurun_out/$R5/stats_headline.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$R5/stats -o b -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu > /root/repo/gpurun_out/$R5/stats.log 2>&1 )
bash tools/tools_pmc.sh $R5 spmm_tiled fetch write tcc sq > gpurun_out/$R5/pmc_headline.json 2>&1
cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/$R5/pmc_paths
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$R5/pmc_paths/$name -o p -- python $R/bench_paths.py > $R/gpurun_out/$R5/pmc_paths/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python $R/tools/summarise_profiles.py $R/gpurun_out/$R5 > $R/gpurun_out/$R5/summary.txt 2>&1
# only summaries travel back (gpurun merges at most 64 MiB): the raw per-dispatch counter tables stay on the box
find $R/gpurun_out/$R5 $R/gpurun_out/pmc_$R5 -name '*counter_collection.csv' -delete
find $R/gpurun_out/$R5 $R/gpurun_out/pmc_$R5 -name '*kernel_trace.csv' -delete
ls $R/gpurun_out/$R5
