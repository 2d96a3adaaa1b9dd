# Round-2 evidence run (one gpurun call): tests of the touched kernels, the bench line, rocprofv3 kernel stats of the same
# command, PMC passes of the headline kernel (each counter group in its own run), the other section-8 rows, the SDDMM
# crossover and the matrix-core counters of the SDDMM tile kernel.  Everything lands under gpurun_out/r02/.
cd /root/repo
mkdir -p gpurun_out/r02 gpurun_out/pmc_sddmm
timeout 300 python bench.py --steps 30 --warmup 3 2>&1 | tail -1 > gpurun_out/r02/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02/stats -o b -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu --no-paths > /root/repo/gpurun_out/r02/stats.log 2>&1 )
bash tools/tools_pmc.sh r02 spmm_tiled fetch write tcc sq sq2 sq3 > gpurun_out/r02/pmc.json 2>&1
timeout 200 python tools/sddmm_crossover.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/sddmm_crossover.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /root/repo/gpurun_out/pmc_sddmm/mfma -o p -- python /root/repo/tools/sddmm_mfma_profile.py > /root/repo/gpurun_out/pmc_sddmm/mfma.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /root/repo/gpurun_out/pmc_sddmm/grbm -o p -- python /root/repo/tools/sddmm_mfma_profile.py > /root/repo/gpurun_out/pmc_sddmm/grbm.log 2>&1
python /root/repo/tools/tools_pmc_parse.py /root/repo/gpurun_out/pmc_sddmm sddmm_mfma > /root/repo/gpurun_out/r02/sddmm_mfma_pmc.json 2>&1
cd /root/repo
timeout 200 python tools/spgemm_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/spgemm_time.txt
ls gpurun_out/r02
