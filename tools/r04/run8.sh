cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench_paths.py --rows A2_default,A6 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['row'], round(d['ms'], 3), 'ms frac', round(d['frac'], 3))
"
