#!/bin/bash
# rocprofv3 PMC passes for the bench's dominant kernel (each counter group in its own run,
# with --kernel-trace only, as gpurun requires).  Output CSVs under gpurun_out/pmc_<tag>/.
# usage: [BENCH_ARGS=--no-tiled] tools_pmc.sh <tag> <kernel-name-substring> [pass ...]   (passes: fetch write tcc sq grbm)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
tag=${1:-r01}; needle=${2:-spmm_csr}; shift; shift
passes=${@:-fetch write}
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag/$name -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-paths $BENCH_ARGS > $R/gpurun_out/pmc_$tag/$name.log 2>&1; }
mkdir -p $R/gpurun_out/pmc_$tag
for p in $passes; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    tcc) run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
    sq) run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU ;;
    grbm) run grbm GRBM_GUI_ACTIVE ;;
    sq2) run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM ;;
    sq3) run sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC ;;
    sq4) run sq4 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT ;;
  esac
done
python $R/tools/tools_pmc_parse.py $R/gpurun_out/pmc_$tag $needle
