# round-5 call 1: the new compact bench line + the reference's small benchmark workloads (baseline before host-side changes)
cd /root/repo
mkdir -p gpurun_out/r05
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_stdout.txt 2> gpurun_out/r05/bench_stderr.txt
tail -1 gpurun_out/r05/bench_stdout.txt > gpurun_out/r05/bench_line.json
wc -c gpurun_out/r05/bench_line.json
timeout 900 python bench_small.py --profile --out gpurun_out/r05/small_baseline.json > gpurun_out/r05/small_baseline.txt 2>&1
tail -60 gpurun_out/r05/small_baseline.txt
