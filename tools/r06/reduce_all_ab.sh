#!/bin/bash
# spamd_reduce_all: pieces (ticket atomics) x loads in flight, one box
cd "$(dirname "$0")/../.."
for v in "" u2 u8; do
  if [ -n "$v" ]; then export SPAMD_LIB=$PWD/sparse_amd/_lib/variants/libsparse_amd_$v.so; else unset SPAMD_LIB; fi
  for p in 128 256 512 1024 2048; do
    echo -n "${v:-u4} "; SPAMD_RA_PIECES=$p python tools/r06/reduce_all_kernel.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
