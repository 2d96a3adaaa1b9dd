#!/usr/bin/env python
"""Verify that hipcc's own code in the tiled SpMM kernels never touches the VGPRs the generated asm owns.

The executor keeps its accumulators (and scratch) in fixed VGPRs across several asm blocks; between those blocks the
compiler's code runs (chunk loop, partial-tile DMA, list offsets).  `amdgpu_num_vgpr(22)` asks the compiler to stay
in v0..v21, but LLVM drops that request when it conflicts with its occupancy bounds, so this script checks the
compiled code instead: every instruction of the spmm_tiled kernels OUTSIDE the `;;#ASMSTART` / `;;#ASMEND` regions
may only use registers below the accumulator block (the first register of TL_CLOB_ACC in the generated include:
v56 for the 35-row geometry).  Scratch registers of the asm (v22..v55) are dead between asm blocks and free for the
compiler; the walking DMA pointer is an in/out operand.

    python tools/check_tiled_regs.py [extra hipcc flags]      exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    inc = open(os.path.join(ROOT, "sparse_amd", "csrc", "spmm_tiled_asm.inc")).read()
    LIMIT = int(re.search(r'#define TL_CLOB_ACC "v(\d+)"', inc).group(1))
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc",
               "-I" + os.path.join(ROOT, "include"), "--save-temps=obj", "-c",
               os.path.join(ROOT, "sparse_amd", "csrc", "spmm_tiled.hip"), "-o", os.path.join(tmp, "t.o")] + sys.argv[1:]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-2000:])
            return 2
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        text = open(os.path.join(tmp, asm[0])).read().split("\n")
    bad, kernel, in_asm, kernels = [], None, False, 0
    reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    for n, line in enumerate(text, 1):
        m = re.match(r"^(_ZN5spamd17spmm_tiled_kernel\w+):", line)
        if m:
            kernel, kernels = m.group(1), kernels + 1
            continue
        if kernel is None:
            continue
        if line.startswith(".Lfunc_end"):      # (not the first s_endpgm: workgroups beyond the grid's useful part return early)
            kernel = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm and not line.lstrip().startswith((";", ".")):
            for a, lo, hi in reg.findall(line.split(";")[0]):
                top = int(a) if a else int(hi)
                if top >= LIMIT:
                    bad.append((kernel, n, line.strip()))
            # an SGPR spilled into a VGPR lane: round 5 had a build that kept a 64-bit group index live across the phase loop,
            # ran out of the SGPRs the asm blocks leave the compiler, spilled (v_writelane / v_readlane around the loop) and
            # faulted intermittently on the GPU - refused here
            if "v_writelane_b32" in line:
                bad.append((kernel, n, "SGPR spill: " + line.strip()))
    print(f"checked {kernels} spmm_tiled kernels: {len(bad)} compiler instruction(s) touch v{LIMIT}+ or spill an SGPR")
    for k, n, l in bad[:20]:
        print(f"  {k[:60]} line {n}: {l}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
