"""Audit of compiler output (hipcc -S --cuda-device-only): per kernel, vector-memory loads whose result is waited for with
`s_waitcnt vmcnt(0)` within a few instructions of their issue ("load then stall": a load under a divergent branch, or a chain of
dependent loads) and how many of them sit inside loops.     python tools/isa_wait_audit.py /tmp/isa/*.s [min_count]"""
import re, sys
files = [a for a in sys.argv[1:] if a.endswith(".s")]
minc = int(sys.argv[-1]) if not sys.argv[-1].endswith(".s") else 3
rows = []
for f in files:
    name, body = None, []
    for line in open(f, errors="replace"):
        m = re.match(r"^(_Z\S+):\s", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB") or "Loop Header" in t:
                body.append(("label", t))
            continue
        body.append(("ins", t))
        if t.startswith("s_endpgm"):
            ins = [x for x in body]
            loads = stalled = inloop = 0
            depth = 0
            for i, (k, t2) in enumerate(ins):
                if k == "label":
                    if "Loop Header" in t2 or "Inner Loop" in t2:
                        depth = 1
                    continue
                if re.match(r"(global|buffer|flat)_load", t2):
                    loads += 1
                    n = 0
                    for k3, t3 in ins[i + 1:i + 14]:
                        if k3 != "ins":
                            continue
                        n += 1
                        if re.match(r"(global|buffer|flat)_(load|store)", t3):
                            break
                        if t3.startswith("s_waitcnt") and "vmcnt(0)" in t3:
                            stalled += 1
                            break
                        if n >= 8:
                            break
            rows.append((stalled, loads, f.split("/")[-1], name))
            name = None
rows.sort(reverse=True)
for st, ld, f, n in rows:
    if st >= minc:
        print(f"{st:4d} of {ld:4d} loads waited at once  {f:22s} {n[:110]}")
