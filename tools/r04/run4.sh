cd /root/repo
for v in "" sdp_wpe4 sdp_nopipe; do
  if [ -z "$v" ]; then lib=""; else lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"; fi
  echo "== ${v:-shipped (pipe, 5 waves)}"
  SPAMD_LIB=$lib timeout 200 python tools/sddmm_cfg4.py 2>&1 | tail -1 | cut -c1-160
done
timeout 600 python -m pytest tests/test_sddmm_gpu.py -x -q -m gpu 2>&1 | tail -2
