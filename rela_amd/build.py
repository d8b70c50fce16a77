"""Builds the native pieces in-tree (no JIT cache): librela_amd.so (HIP kernels + C ABI, hipcc,
gfx950) and, on request, the pybind11 module `rela` that mirrors rela/pybind.cc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "librela_amd.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the replay bookkeeping must round exactly like the reference's scalar code
# (no silent FMA); kernels that want FMA/MFMA ask for it explicitly.
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
             "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_native(verbose=False, force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "rela_amd.h")]
    objs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s)[:-4] + ".o")
        if force or _newer(o, deps):
            cmd = [HIPCC] + HIP_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_native(verbose=True, force="--force" in sys.argv)
    print(LIB)
