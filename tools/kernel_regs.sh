#!/bin/bash
# registers / spills of every kernel of one csrc/*.hip file:  bash tools/kernel_regs.sh spmm_stream [extra hipcc flags]
cd "$(dirname "$0")/../sparse_amd/csrc" || exit 1
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable \
  -Rpass-analysis=kernel-resource-usage "$@" -c $f.hip -o /tmp/$f.regs.o 2>&1 |
  grep -E "error|Function Name|VGPRs:|VGPRs Spill|SGPRs Spill" | paste - - - - |
  sed -E 's/.*Function Name: ([^ ]*) .*VGPRs: ([0-9]*).*SGPRs Spill: ([0-9]*).*VGPRs Spill: ([0-9]*).*/\1 vgpr=\2 sspill=\3 vspill=\4/'
