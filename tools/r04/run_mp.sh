#!/bin/bash
# merge-path tile shapes at 10^8 + 10^8 items (A7_1e8 rows) and at config 1
for v in ${VARIANTS:-"" mp_t512v4 mp_t1024v4 mp_t256v8 mp_t256v4}; do
  if [ -n "$v" ]; then export SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_$v.so; else unset SPAMD_LIB; fi
  echo "== ${v:-default}"
  python bench_paths.py --rows A7 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(' ', r['row'], round(r['ms'], 4))"
done
