# rocprofv3 PMC passes for the sampled SDDMM kernel on config 4 (bf16): one counter group per run, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_sddmm; mkdir -p $R/gpurun_out/pmc_sddmm
cat > /tmp/sddmm_once.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
Ms = 100_000
s = sp.random((Ms, Ms), nnz=10_000_000, random_state=3, dtype=np.float32, idx_dtype=np.int32)
a = torch.rand((Ms, 256), device="cuda").to(torch.bfloat16); bt = torch.rand((Ms, 256), device="cuda").to(torch.bfloat16)
for _ in range(3): r = K.sddmm_coo(s.coords, s.data, a, bt)
torch.cuda.synchronize()
PY
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_sddmm/$name -o p -- python /tmp/sddmm_once.py > $R/gpurun_out/pmc_sddmm/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run grbm GRBM_GUI_ACTIVE
python $R/tools/tools_pmc_parse.py $R/gpurun_out/pmc_sddmm sddmm_rowcache_kernel
tail -3 $R/gpurun_out/pmc_sddmm/tcp.log $R/gpurun_out/pmc_sddmm/ta.log
