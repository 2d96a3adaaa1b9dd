"""Static check of spmm_stream.hip's hand-counted loads (run by sparse_amd/csrc/build.py's caller on demand and by
tests/test_host_logic.py): in every instantiation of spmm_stream_kernel, no instruction may touch a register of a ring
buffer between the buffer's `global_load_dwordx4` (inline assembly; invisible to the compiler's waitcnt insertion) and the
hand-written `s_waitcnt vmcnt(n)` that hands it over.  The kernel's ISA is walked in LAYOUT order with one state per
buffer (free -> in flight at its load -> ready at its wait -> ...); a block the compiler moved out of line is checked
under the state of the place it was moved to, which can only produce false alarms, not misses, for the main path the
loads live on.  Also refuses spilled registers (a spill of a buffer register would be such a touch).

    python tools/check_stream_regs.py [extra hipcc flags]      exit status 1 on a finding
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sparse_amd", "csrc", "spmm_stream.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-w",
         "--cuda-device-only", "-S"]


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def check_kernel(name, lines):
    findings = []
    # pass 1: the ring's registers = destinations of the inline-assembly loads; buffers = groups of loads between waits
    in_asm = False
    loads = []     # (line no, dst regs)
    for n, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        elif in_asm and "global_load_dword" in ln:
            loads.append((n, frozenset(regs_of(ln.split(",")[0]))))
    if not loads:
        return ["no inline-assembly loads found"]
    ring = set().union(*(r for _, r in loads))
    # a buffer = the destination registers of one run of consecutive asm loads
    state = {}     # reg -> "flight" | "ready"
    pending = []   # load groups in issue order: list of frozenset
    in_asm = False
    for n, ln in enumerate(lines):
        s = ln.strip()
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        if in_asm and "global_load_dword" in s:
            dst = regs_of(s.split(",")[0])
            addr = regs_of(",".join(s.split(",")[1:]))
            bad = [r for r in addr if state.get(r) == "flight"]
            if bad:
                findings.append(f"{name}: line {n}: load address in a register in flight: {s}")
            for r in dst:
                state[r] = "flight"
            pending.append(dst)
            continue
        if in_asm and s.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m:
                keep = int(m.group(1))
                while len(pending) > keep:
                    for r in pending.pop(0):
                        state[r] = "ready"
            continue
        m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", s)
        if m and int(m.group(1)) == 0:   # a compiler-made drain: everything has landed
            for grp in pending:
                for r in grp:
                    state[r] = "ready"
            pending = []
            continue
        touched = [r for r in regs_of(s) if r in ring and state.get(r) == "flight"]
        if touched:
            findings.append(f"{name}: line {n}: touches v{sorted(touched)} in flight: {s}")
    return findings


def main(extra):
    asm = subprocess.run(["hipcc", *FLAGS, *extra, SRC, "-o", "-"], capture_output=True, text=True)
    if asm.returncode != 0:
        print(asm.stderr)
        return 1
    text = asm.stdout
    findings = []
    kernels = 0
    for m in re.finditer(r"^(_ZN5spamd18spmm_stream_kernel\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M):
        kernels += 1
        findings += check_kernel(m.group(1), m.group(2).split("\n"))
    for m in re.finditer(r"\.(?:vgpr|sgpr)_spill_count:\s+(\d+)", text):
        pass
    spills = re.findall(r"^\s*; (?:ScratchSize|NumVgprs|.*[Ss]pill.*): .*$", text, flags=re.M)
    for name, n in re.findall(r"\.name:\s+(_ZN5spamd18spmm_stream_kernel\w+).*?\.vgpr_spill_count:\s+(\d+)", text, flags=re.S):
        if int(n):
            findings.append(f"{name}: {n} spilled VGPRs")
    print(f"{kernels} instantiations checked, {len(findings)} findings")
    for f in findings[:40]:
        print("  " + f)
    return 1 if findings or not kernels else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
