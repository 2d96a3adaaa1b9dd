"""Print the top rows of a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
