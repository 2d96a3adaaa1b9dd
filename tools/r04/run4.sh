cd /root/repo
for v in "" sdp_ch2 sdp_ch4; do
  if [ -z "$v" ]; then lib=""; else lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"; fi
  echo -n "== ${v:-shipped}: "
  SPAMD_LIB=$lib timeout 200 python tools/sddmm_cfg4.py 2>&1 | tail -1 | cut -c1-75
done
