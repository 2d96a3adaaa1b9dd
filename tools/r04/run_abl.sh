#!/bin/bash
export SPAMD_MERGE_STREAM=0
for v in "" abl1 abl2 abl3; do
  if [ -n "$v" ]; then export SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_$v.so; else unset SPAMD_LIB; fi
  echo "== ${v:-default}"; python tools/r04/merge_stream_check.py 2>&1 | tail -2
done
