cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_spg; mkdir -p $R/gpurun_out/pmc_spg
cat > /tmp/spg_once.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
g = sp.random((100_000, 100_000), density=1e-3, random_state=7, dtype=np.float32, idx_dtype=np.int32, format="gcxs", compressed_axes=(0,))
for _ in range(2): c = g @ g
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_spg/sq -o p -- python /tmp/spg_once.py > $R/gpurun_out/pmc_spg/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_spg/sq2 -o p -- python /tmp/spg_once.py > $R/gpurun_out/pmc_spg/sq2.log 2>&1
python $R/tools/tools_pmc_parse.py $R/gpurun_out/pmc_spg rowrank_kernel
