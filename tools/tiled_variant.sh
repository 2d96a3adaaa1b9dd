#!/bin/bash
# Build libsparse_amd with an experimental tiled-SpMM geometry:  tools/tiled_variant.sh NAME TL_RG=.. TL_WAVES=.. [more env]
# -> sparse_amd/_lib/variants/libsparse_amd_NAME.so ; the generated include is restored to the shipped geometry afterwards.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m sparse_amd.csrc.build > /dev/null
mkdir -p sparse_amd/_lib/variants build_variant
cp sparse_amd/csrc/spmm_tiled_asm.inc build_variant/shipped_asm.inc
env "$@" python tools/gen_tiled_asm.py
obj=build_variant/${name}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Iinclude -DSPAMD_TL_EXPERIMENTAL -DSPAMD_TUNING -c sparse_amd/csrc/spmm_tiled.hip -o $obj 2> build_variant/${name}.log || { cp build_variant/shipped_asm.inc sparse_amd/csrc/spmm_tiled_asm.inc; grep -m5 error build_variant/${name}.log; exit 1; }
cp build_variant/shipped_asm.inc sparse_amd/csrc/spmm_tiled_asm.inc
touch -r build_variant/shipped_asm.inc sparse_amd/csrc/spmm_tiled_asm.inc
objs=$(ls sparse_amd/_lib/obj/*.o | grep -v "/spmm_tiled\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o sparse_amd/_lib/variants/libsparse_amd_${name}.so $objs $obj
echo sparse_amd/_lib/variants/libsparse_amd_${name}.so
