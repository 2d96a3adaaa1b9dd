#!/bin/bash
# Timing ablations of the SpGEMM row kernel: builds libsparse_amd.so with -DSPG_ABL=<n> (1 = stop after
# the expansion, 2 = after the bucket counts and their scan, 3 = everything but the emission; WRONG RESULTS by design),
# to be followed on the GPU box by
#   rocprofv3 --kernel-trace --stats -- python tools/spgemm_abl_time.py
# and by a rebuild without the flag (`touch sparse_amd/csrc/spgemm_rows.hip; python -m sparse_amd.csrc.build`).
# usage: tools/spgemm_ablate.sh <n>
cd "$(dirname "$0")/.."
touch sparse_amd/csrc/spgemm_rows.hip
python - <<PY
import sys
sys.path.insert(0, ".")
from sparse_amd.csrc import build as b
b.CXXFLAGS[:0] = ["-DSPG_ABL=$1"]
print(b.build())
PY
