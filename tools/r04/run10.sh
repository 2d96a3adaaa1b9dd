cd /root/repo
for v in bmk_noload; do
  lib="sparse_amd/_lib/variants/libsparse_amd_$v.so"
  echo "== $v"
  SPAMD_LIB=$lib timeout 300 python tools/r04/spgemm_ab.py 3 2>&1 | grep "bitmap" | cut -c1-100
done
