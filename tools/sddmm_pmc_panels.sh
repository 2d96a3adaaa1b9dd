# TCC hit/miss and fetch size of the sampled SDDMM kernel in column-panel order (config 4, bf16), one counter group per run.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_sddmm_panels; mkdir -p $R/gpurun_out/pmc_sddmm_panels
cat > /tmp/sddmm_once.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import sparse_amd as sp
from sparse_amd import _kernels as K
Ms = 100_000
s = sp.random((Ms, Ms), nnz=10_000_000, random_state=3, dtype=np.float32, idx_dtype=np.int32)
a = torch.rand((Ms, 256), device="cuda").to(torch.bfloat16); bt = torch.rand((Ms, 256), device="cuda").to(torch.bfloat16)
plan = K.sddmm_panels(s.coords, s.shape, K.sddmm_panel_width(bt))
for _ in range(3): r = K.sddmm_coo(s.coords, s.data, a, bt, panels=plan)
torch.cuda.synchronize()
PY
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_sddmm_panels/$name -o p -- python /tmp/sddmm_once.py > $R/gpurun_out/pmc_sddmm_panels/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
python $R/tools/tools_pmc_parse.py $R/gpurun_out/pmc_sddmm_panels sddmm_rowcache_kernel | grep -v dispatches | tr -d '\n{}' 
