#!/bin/bash
# Build an experimental variant of libsparse_amd.so with extra -D flags for ONE source file:
#   tools/build_variant.sh NAME file.hip -DSPAMD_TL_KB=64 -DSPAMD_TL_NBUF=4
# -> sparse_amd/_lib/variants/libsparse_amd_NAME.so   (use with SPAMD_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m sparse_amd.csrc.build > /dev/null
mkdir -p sparse_amd/_lib/variants build_variant
obj=build_variant/${name}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Iinclude "$@" -c sparse_amd/csrc/$src -o $obj
objs=$(ls sparse_amd/_lib/obj/*.o 2>/dev/null | grep -v "/${src%.hip}\.o" || true)
if [ -z "$objs" ]; then echo "object dir not found"; exit 1; fi
hipcc --offload-arch=gfx950 -shared -fPIC -o sparse_amd/_lib/variants/libsparse_amd_${name}.so $objs $obj
echo sparse_amd/_lib/variants/libsparse_amd_${name}.so
