cd /root/repo
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_nd_gpu.py tests/test_fullsize_properties_gpu.py tests/test_round4_gpu.py -x -q -m gpu -k "elem or merge or union or add or config1 or index or where or golden" 2>&1 | tail -2
timeout 600 python bench_paths.py --rows A7 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['row'], round(d['ms'], 3), 'ms frac', round(d['frac'], 3))
"
