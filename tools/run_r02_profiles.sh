cd /root/repo
timeout 200 python -m pytest tests/test_sddmm_gpu.py -x -q 2>&1 | tail -3
timeout 300 python tools/sddmm_crossover.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sddmm_crossover.txt
mkdir -p gpurun_out/r02; bash tools/tools_pmc.sh r02 spmm_tiled fetch write tcc sq sq2 sq3 > gpurun_out/r02/pmc.json 2>&1
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/pmc_sddmm
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /root/repo/gpurun_out/pmc_sddmm/mfma -o p -- python /root/repo/tools/sddmm_mfma_profile.py > /root/repo/gpurun_out/pmc_sddmm/mfma.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /root/repo/gpurun_out/pmc_sddmm/grbm -o p -- python /root/repo/tools/sddmm_mfma_profile.py > /root/repo/gpurun_out/pmc_sddmm/grbm.log 2>&1
python /root/repo/tools/tools_pmc_parse.py /root/repo/gpurun_out/pmc_sddmm sddmm_mfma > /root/repo/gpurun_out/pmc_sddmm/summary.json 2>&1
tail -5 /root/repo/gpurun_out/pmc_sddmm/mfma.log
