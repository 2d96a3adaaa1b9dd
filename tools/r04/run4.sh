cd /root/repo
mkdir -p gpurun_out/r04d
for v in "" sdp_abl1 sdp_abl2 sdp_blk512 sdp_unr2; do
  if [ -z "$v" ]; then lib=""; else lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"; fi
  echo "== ${v:-shipped}" >> gpurun_out/r04d/sddmm.log
  SPAMD_LIB=$lib timeout 200 python tools/sddmm_cfg4.py 2>&1 | tail -1 >> gpurun_out/r04d/sddmm.log
done
cat gpurun_out/r04d/sddmm.log
