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
# The checker (LIB) is built WITHOUT -march=native: the .so built in the container travels to the GPU box and must run
# on its host CPU too.  bench.py's cpu_baseline leg calls build(native=True) ON the box it times: a second library
# (LIB_NATIVE, never shipped) with -march=native, as BASELINE.md section 4.1 prescribes.  -ffp-contract=off in both, so
# the results are bit-identical to each other and to the reference's separate multiply and add.
CFLAGS = ["-O3", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall"]
LIB_NATIVE = os.path.join(OUT_DIR, "liboracle_native.so")


def is_stale(lib=LIB):
    return (not os.path.exists(lib)) or os.path.getmtime(lib) < max(
        os.path.getmtime(SRC), os.path.getmtime(os.path.abspath(__file__)))


def build(force=False, native=False):
    if native:
        return _build_native(force)
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


def _build_native(force=False):
    """-march=native build for THIS host (bench.py's cpu_baseline); rebuilt whenever the host CPU differs."""
    import platform

    tag = os.path.join(OUT_DIR, "native.host")
    host = platform.processor() + "|" + (open("/proc/cpuinfo").read().split("model name", 2)[1].split("\n", 1)[0]
                                           if os.path.exists("/proc/cpuinfo") else "")
    same_host = os.path.exists(tag) and open(tag).read() == host
    if not force and same_host and not is_stale(LIB_NATIVE):
        return LIB_NATIVE
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("gcc not found")
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [cc, *CFLAGS, "-march=native", SRC, "-o", LIB_NATIVE + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{' '.join(cmd)}\n{r.stderr}")
    os.replace(LIB_NATIVE + ".tmp", LIB_NATIVE)
    with open(tag, "w") as f:
        f.write(host)
    return LIB_NATIVE


if __name__ == "__main__":
    print(build(force=True))
