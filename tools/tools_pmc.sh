#!/bin/bash
# rocprofv3 PMC passes for the bench's dominant kernel (each counter group in its own run,
# with --kernel-trace only, as gpurun requires).  Output CSVs under gpurun_out/pmc_<tag>/.
# usage: tools_pmc.sh <tag> <kernel-name-substring> [pass ...]   (passes: fetch write tcc sq grbm)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
tag=${1:-r01}; needle=${2:-spmm_csr}; shift; shift
passes=${@:-fetch write}
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag/$name -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/pmc_$tag/$name.log 2>&1; }
mkdir -p $R/gpurun_out/pmc_$tag
for p in $passes; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    tcc) run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
    sq) run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU ;;
    grbm) run grbm GRBM_GUI_ACTIVE ;;
  esac
done
python $R/tools/tools_pmc_parse.py $R/gpurun_out/pmc_$tag $needle
