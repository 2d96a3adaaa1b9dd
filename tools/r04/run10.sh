cd /root/repo
for v in "" bmk_nolb; do
  if [ -z "$v" ]; then lib=""; else lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"; fi
  echo "== ${v:-shipped}"
  SPAMD_LIB=$lib timeout 300 python tools/r04/spgemm_ab.py 3 2>&1 | grep "bitmap\|identical"
done
