import json, sys
keys = sys.argv[2:] or None
for line in open(sys.argv[1]):
    if line.startswith('{"row"'):
        d = json.loads(line)
        out = {}
        for k, v in d.items():
            if k in ("workload", "cpu_baseline", "pmc_null_reason", "api_note", "row", "algorithmic_bytes", "GBps", "pmc_bytes", "pmc_over_algorithmic"):
                continue
            out[k] = round(v, 4) if isinstance(v, float) else v
        cb = d.get("cpu_baseline") or {}
        for k in ("max_err_over_sum_abs_terms", "max_rel_err", "keys_bit_exact", "indices_bit_exact"):
            if k in cb:
                out["cpu:" + k] = cb[k]
        print(d["row"], out)
