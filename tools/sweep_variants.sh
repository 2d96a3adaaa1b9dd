cd /root/repo
for lib in "" sparse_amd/_lib/variants/libsparse_amd_TL_*.so; do
  SPAMD_LIB=$lib timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu --no-paths 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('${lib:-default}', 'kernel_ms', round(d['roofline']['kernel_ms'], 4), 'ms_per_step', round(d['ms_per_step'], 4))
"
done
