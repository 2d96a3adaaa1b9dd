"""Build libsparse_amd.so (HIP, gfx950) in-tree with hipcc.

    python -m sparse_amd.csrc.build            # build if stale
    python -m sparse_amd.csrc.build --force

The shared library lands in sparse_amd/_lib/libsparse_amd.so (git-ignored, but shipped to
the GPU box by gpurun).  hipcc cross-compiles for gfx950 without a GPU present.
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT_DIR = os.path.join(PKG, "_lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libsparse_amd.so")
ARCH = "gfx950"
CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libsparse_amd.so)")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _deps():
    return sorted(glob.glob(os.path.join(HERE, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "*.inc"))) + [
        os.path.join(os.path.dirname(PKG), "include", "sparse_amd.h"), os.path.abspath(__file__)]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources() + _deps())


def _own_deps(src, obj):
    """The files `src` really includes, from the compiler's dependency file of the last build (-MD); every header and
    .inc of csrc/ when there is none yet."""
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep):
        return _deps()
    words = open(dep).read().replace("\\\n", " ").split()
    files = [w for w in words[1:] if not w.startswith("/opt/") and not w.startswith("/usr/")]
    if any(not os.path.exists(f) for f in files):
        return _deps()
    return files + [os.path.abspath(__file__)]


# Kernels that share their register file with hand-written assembly (spmm_tiled: accumulators in a fixed VGPR block, VGPR
# index mode around the list loop) must not have ANY register spilled by the compiler: a round-5 build whose SGPR pressure
# made hipcc park a value in a VGPR lane (v_writelane / v_readlane around the phase loop) faulted on the GPU one product in
# ten.  The build fails on such an object instead of shipping it (tools/check_tiled_regs.py checks the same and more).
NO_SPILL = {"spmm_tiled.hip": "spmm_tiled_kernel"}


def _spills(remarks, needle):
    """[(function, sgpr spills, vgpr spills)] with a non-zero count, from -Rpass-analysis=kernel-resource-usage remarks"""
    out, name, sg = [], None, 0
    for ln in remarks.splitlines():
        if "Function Name:" in ln:
            name = ln.split("Function Name:")[1].split()[0]
        elif "SGPRs Spill:" in ln:
            sg = int(ln.split("SGPRs Spill:")[1].split()[0])
        elif "VGPRs Spill:" in ln and name is not None:
            vg = int(ln.split("VGPRs Spill:")[1].split()[0])
            if needle in name and (sg or vg):
                out.append((name, sg, vg))
    return out


def _compile(src, force):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    newest = max(os.path.getmtime(p) for p in [src] + _own_deps(src, obj))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, ""
    needle = NO_SPILL.get(os.path.basename(src))
    extra = ["-Rpass-analysis=kernel-resource-usage"] if needle else []
    cmd = [_hipcc(), *CXXFLAGS, *extra, "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if needle:
        bad = _spills(r.stderr, needle)
        if bad:
            os.remove(obj)
            raise RuntimeError(f"{os.path.basename(src)}: the compiler spilled registers in " +
                               ", ".join(f"{n[:60]} ({s} SGPR, {v} VGPR)" for n, s, v in bad[:4]) +
                               " - refused (see NO_SPILL in sparse_amd/csrc/build.py)")
        return obj, "\n".join(ln for ln in r.stderr.splitlines() if "-Rpass-analysis" not in ln and "remark:" not in ln)
    return obj, r.stderr


def build(force=False, verbose=False, jobs=None):
    """Compile every csrc/*.hip for gfx950 and link libsparse_amd.so. Returns its path."""
    if not force and not is_stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    jobs = jobs or min(len(srcs), os.cpu_count() or 1)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    if verbose:
        for _, w in results:
            if w.strip():
                print(w, file=sys.stderr)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print(p)
