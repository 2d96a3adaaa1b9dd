"""Summarise rocprofv3 PMC CSVs: per-counter average per dispatch of kernels matching a name."""
import csv
import glob
import json
import os
import sys

root, needle = sys.argv[1], sys.argv[2]
out = {}
for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if needle not in row.get("Kernel_Name", ""):
                continue
            name = row["Counter_Name"]
            acc.setdefault(name, {})
            d = row["Dispatch_Id"]
            acc[name][d] = acc[name].get(d, 0.0) + float(row["Counter_Value"])
    for name, per in acc.items():
        vals = list(per.values())
        out[name] = {"avg_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
print(json.dumps(out, indent=1))
with open(os.path.join(root, "summary.json"), "w") as f:
    json.dump(out, f, indent=1)
