"""Register / scratch / occupancy table of every kernel of libsparse_amd.so, from hipcc's own resource remarks (no GPU needed):
    python tools/kernel_resources.py [out.txt]
Compiles every csrc/*.hip for gfx950 device-only with -Rpass-analysis=kernel-resource-usage and lists, per kernel instance:
VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, waves per SIMD.  Kernels that spill are listed first."""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{ROOT}/sparse_amd/csrc",
         "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]


def one(src):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["hipcc", *FLAGS, "-c", src, "-o", os.path.join(td, "x.co")], capture_output=True, text=True)
    rows = []
    for b in r.stderr.split("Function Name: ")[1:]:
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        name = subprocess.run(["c++filt", b.split(" ")[0]], capture_output=True, text=True).stdout.strip()
        rows.append((g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(" VGPRs"), g("AGPRs"),
                     g(r"Occupancy \[waves/SIMD\]"), os.path.basename(src), re.sub(r"\(.*", "", name)[:110]))
    return rows


srcs = sorted(glob.glob(f"{ROOT}/sparse_amd/csrc/*.hip"))
with ThreadPoolExecutor(max_workers=8) as ex:
    rows = [r for rs in ex.map(one, srcs) for r in rs]
rows.sort(key=lambda r: (-r[0], r[6], r[7]))
lines = [f"{len(rows)} kernel instances in {len(srcs)} files; {sum(1 for r in rows if r[0])} spill VGPRs",
         "vgpr_spill sgpr_spill scratch_B vgprs agprs waves/SIMD file kernel"]
lines += [f"{r[0]:5d} {r[1]:5d} {r[2]:5d} {r[3]:4d} {r[4]:4d} {r[5]:3d}  {r[6]:20s} {r[7]}" for r in rows]
out = "\n".join(lines)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out + "\n")
print("\n".join(lines[:40]))
