"""Build librsx.so (the C-ABI HIP library) in-tree for gfx950.

    python -m recsys_pytorch_amd.build [--force] [--dev]

--dev additionally builds librsx_dev.so with -DRSX_ABLATE (development write/load switches for
tools/ablate*.py; load it with RSX_LIB=.../librsx_dev.so).  The product library never has them.

hipcc cross-compiles without a GPU.  The .so lands next to this file so it
travels with the source tree (git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librsx.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


DEV_LIB = os.path.join(HERE, "librsx_dev.so")


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(HERE, "..", "include", "rsx.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, dev=False):
    lib = DEV_LIB if dev else LIB
    if not force and not _stale(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build", "dev" if dev else "")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *(["-DRSX_ABLATE"] if dev else []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs]
    subprocess.check_call(cmd)
    # a shared library links with undefined symbols: load it once, so that a kernel whose host stub went missing fails the BUILD
    # (no GPU needed for dlopen)
    import ctypes
    ctypes.CDLL(lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, dev=True))
