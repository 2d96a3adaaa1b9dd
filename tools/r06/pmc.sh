#!/bin/bash
# rocprofv3 counter passes over any command (each counter group in its own run, --kernel-trace only):
#   bash tools/r06/pmc.sh <tag> <kernel-name-substring> "<passes>" <command ...>      passes: fetch write tcc sq sq2 sq3 stats
# summaries: gpurun_out/pmc_<tag>/summary.json (+ stats_<tag>.csv for `stats`)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
tag=$1; needle=$2; passes=$3; shift; shift; shift
mkdir -p $R/gpurun_out/pmc_$tag
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag/$name -o p -- "${CMD[@]}" > $R/gpurun_out/pmc_$tag/$name.log 2>&1; }
CMD=("$@")
for p in $passes; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    tcc) run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
    sq) run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU ;;
    sq2) run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM ;;
    sq3) run sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC ;;
    grbm) run grbm GRBM_GUI_ACTIVE ;;
    stats) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_$tag/stats -o s -- "${CMD[@]}" > $R/gpurun_out/pmc_$tag/stats.log 2>&1
           f=$(find $R/gpurun_out/pmc_$tag/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/pmc_$tag/kernel_stats.csv && head -8 $f ;;
  esac
done
python $R/tools/tools_pmc_parse.py $R/gpurun_out/pmc_$tag $needle
find $R/gpurun_out/pmc_$tag -name '*counter_collection.csv' -delete
find $R/gpurun_out/pmc_$tag -name '*kernel_trace.csv' -delete
find $R/gpurun_out/pmc_$tag -name '*agent_info.csv' -delete
