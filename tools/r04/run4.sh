cd /root/repo
mkdir -p gpurun_out/r04d
rm -f gpurun_out/r04d/sddmm.log
for v in "" sdp_ch2 sdp_ch4 sdp_ch8; do
  if [ -z "$v" ]; then lib=""; else lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"; fi
  echo "== ${v:-shipped}" >> gpurun_out/r04d/sddmm.log
  SPAMD_LIB=$lib timeout 200 python tools/sddmm_cfg4.py 2>&1 | tail -1 >> gpurun_out/r04d/sddmm.log
done
cat gpurun_out/r04d/sddmm.log
SPAMD_LIB=sparse_amd/_lib/variants/libsparse_amd_sdp_ch4.so timeout 600 python -m pytest tests/test_sddmm_gpu.py -x -q -m gpu 2>&1 | tail -2
