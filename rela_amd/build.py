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


PYBIND_DIR = os.path.join(PKG, "pybind")


def build_pybind(verbose=False, force=False):
    """g++ build of the `rela` and `synth_atari` extension modules (torch's own vendored pybind11
    headers, so tensors cast through torch's type registry).  ~2 minutes per module."""
    import sysconfig

    import torch

    tdir = os.path.dirname(torch.__file__)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PYBIND_DIR, "include"),
           "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I" + sysconfig.get_paths()["include"]]
    flags = ["-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-D_GLIBCXX_USE_CXX11_ABI=1", "-w"]
    libs = ["-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
            "-Wl,-rpath," + os.path.join(tdir, "lib")]
    hdrs = glob.glob(os.path.join(PYBIND_DIR, "include", "rela", "*.h")) + [os.path.join(ROOT, "include", "rela_amd.h")]
    jobs = []
    outs = []
    for name, src, extra in (("rela", "rela_module.cc", ["-L" + PKG, "-lrela_amd", "-Wl,-rpath,$ORIGIN/.."]),
                             ("synth_atari", "synth_atari.cc", [])):
        out = os.path.join(PYBIND_DIR, name + ext)
        outs.append(out)
        srcp = os.path.join(PYBIND_DIR, src)
        if force or _newer(out, [srcp] + hdrs):
            cmd = ["g++"] + flags + inc + [srcp, "-o", out] + libs + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append(subprocess.Popen(cmd))
    for j in jobs:
        if j.wait() != 0:
            raise RuntimeError("pybind build failed")
    return outs


if __name__ == "__main__":
    build_native(verbose=True, force="--force" in sys.argv)
    print(LIB)
    if "--pybind" in sys.argv:
        print(build_pybind(verbose=True, force="--force" in sys.argv))
