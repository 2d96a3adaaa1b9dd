"""per-kernel averages of a rocprofv3 kernel_stats.csv:  python tools/r06/kstat.py <csv> [name-substring]"""
import csv, sys
needle = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if needle in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f}  max {float(r["MaxNs"]) / 1e3:9.1f}')
