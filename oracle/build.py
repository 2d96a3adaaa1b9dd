"""Build the CPU oracle (oracle/oracle.c -> oracle/_build/liboracle.so) with gcc.

TEST INFRASTRUCTURE: the oracle is the parity checker and bench.py's cpu_baseline leg;
sparse_amd never loads it.  -ffp-contract=off keeps multiply and add separate, as the
reference's (numba/LLVM, no fast-math) loops do.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
SRC = os.path.join(HERE, "oracle.c")
# no -march=native: the .so built in the container must run on the GPU box's host CPU too
CFLAGS = ["-O3", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall"]


def is_stale():
    return (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(
        os.path.getmtime(SRC), os.path.getmtime(os.path.abspath(__file__)))


def build(force=False):
    if not force and not is_stale():
        return LIB
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("gcc not found")
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [cc, *CFLAGS, SRC, "-o", LIB + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{' '.join(cmd)}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
