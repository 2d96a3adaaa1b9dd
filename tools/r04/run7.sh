cd /root/repo
timeout 600 python -m pytest tests/test_round4_gpu.py -x -q -m gpu -k "int32" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_spmm_tiled_gpu.py tests/test_spmm_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench_paths.py --rows A1_shapes_int32,A1_first,A2_default,A3_coo 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['row'], round(d['ms'], 3), 'ms frac', round(d['frac'], 3))
"
