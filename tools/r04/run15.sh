cd /root/repo
for v in "" csxwide; do
  if [ -z "$v" ]; then lib=""; else lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"; fi
  echo "== ${v:-shipped}"
  SPAMD_LIB=$lib timeout 300 python tools/r04/csx_time.py 2>&1 | tail -2
done
