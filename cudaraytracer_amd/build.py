"""Builds the native pieces in-tree.

  cudaraytracer_amd/librt_gfx950.so  HIP kernels (gfx950) + C-ABI + C++ host API
  oracle/liboracle.so                CPU restatement used by the tests only

Both are plain compiler invocations (hipcc cross-compiles without a GPU).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cudaraytracer_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "librt_gfx950.so")
ORACLE_SRC = os.path.join(ROOT, "oracle", "oracle.cpp")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No FMA contraction, IEEE divide / sqrt (hipcc default), denormals kept: the
# kernel's fp32 decisions must equal a host evaluation of the same formulas.
DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC"]
HOST_FLAGS = ["-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources():
    out = []
    for base, _dirs, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h")):
                out.append(os.path.join(base, f))
    out.append(os.path.join(ROOT, "include", "rt_capi.h"))
    return out


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_library(force=False, extra_device_flags=(), out=None, tag=""):
    """Builds the library; `out` / `tag` / `extra_device_flags` produce a tuning variant beside the default build."""
    global LIB
    if out is None:
        out = LIB
    if not force and not _newer(out, _sources()):
        return out
    objdir = os.path.join(PKG, "build" + tag)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    kern = os.path.join(CSRC, "kernels", "rt_kernels.hip")
    o = os.path.join(objdir, "rt_kernels.o")
    _run([HIPCC, *DEVICE_FLAGS, *extra_device_flags, "-c", kern, "-o", o])
    objs.append(o)
    for rel in ("rt_capi.cpp", os.path.join("host", "harness.cpp"), os.path.join("host", "host_selftest.cpp")):
        src = os.path.join(CSRC, rel)
        o = os.path.join(objdir, os.path.basename(rel).replace(".cpp", ".o"))
        _run([HIPCC, "-x", "c++", *HOST_FLAGS, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c", src, "-o", o])
        objs.append(o)
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-lz"])
    return out


def build_oracle(force=False):
    if not force and not _newer(ORACLE_LIB, [ORACLE_SRC, os.path.join(ROOT, "include", "rt_capi.h")]):
        return ORACLE_LIB
    _run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-function",
          "-o", ORACLE_LIB, ORACLE_SRC])
    return ORACLE_LIB


def build_all(force=False):
    build_library(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
