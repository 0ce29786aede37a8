"""Builds the native pieces in-tree.

  cudaraytracer_amd/librt_gfx950.so  HIP kernels (gfx950) + C-ABI: exports include/rt_capi.h and nothing else
  cudaraytracer_amd/librt_host.so    C++ host API (csrc/host) behind flat rth_* entry points: scene generators,
                                     experiment reader, OBJ / PNG readers, self-test; calls the C-ABI library
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
HOST_LIB = os.path.join(PKG, "librt_host.so")
ORACLE_SRC = os.path.join(ROOT, "oracle", "oracle.cpp")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
ORACLE_O3_LIB = os.path.join(ROOT, "oracle", "liboracle_o3.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No FMA contraction, IEEE divide / sqrt (hipcc default), denormals kept: the
# kernel's fp32 decisions must equal a host evaluation of the same formulas.
# No SLP vectorisation: packed fp32 instructions issue at half the rate of plain ones on gfx950, so pairs gain nothing
# and the moves that line their operands up cost issue slots and registers (88 -> 76 VGPRs in the trace kernel).
DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-std=c++17", "-fPIC"]
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
    headers = [f for f in _sources() if f.endswith(".h")]

    def compile_one(src, obj, cmd):
        # per-object rebuild: the source itself or any header changed (a forced build recompiles everything)
        if force or _newer(obj, [src] + headers):
            _run(cmd + ["-c", src, "-o", obj])
        objs.append(obj)

    compile_one(os.path.join(CSRC, "kernels", "rt_kernels.hip"), os.path.join(objdir, "rt_kernels.o"), [HIPCC, *DEVICE_FLAGS, *extra_device_flags])
    # the BVH build and tile kernels take no tuning flags; every variant links the default build's objects
    bvh_obj_dir = os.path.join(PKG, "build")
    os.makedirs(bvh_obj_dir, exist_ok=True)
    bvh_force, force = force, (force and not tag)
    compile_one(os.path.join(CSRC, "kernels", "rt_bvh_build.hip"), os.path.join(bvh_obj_dir, "rt_bvh_build.o"), [HIPCC, *DEVICE_FLAGS])
    compile_one(os.path.join(CSRC, "kernels", "rt_tiles.hip"), os.path.join(bvh_obj_dir, "rt_tiles.o"), [HIPCC, *DEVICE_FLAGS])
    force = bvh_force
    # macro definitions of a tuning variant reach the host side too (the two share rt_device_format.h)
    host_cmd = [HIPCC, "-x", "c++", *HOST_FLAGS, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", *[f for f in extra_device_flags if f.startswith("-D")]]
    compile_one(os.path.join(CSRC, "rt_capi.cpp"), os.path.join(objdir, "rt_capi.o"), host_cmd)
    compile_one(os.path.join(CSRC, "rt_multi.cpp"), os.path.join(objdir, "rt_multi.o"), host_cmd)
    # only the entry points of include/rt_capi.h leave the library
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "rt_capi.map"), "-o", out, *objs])
    return out


def build_host_library(force=False):
    """The C++ host classes (csrc/host) behind flat entry points, for Python: links against the C-ABI library."""
    srcs = [os.path.join(CSRC, "host", "harness.cpp"), os.path.join(CSRC, "host", "host_selftest.cpp")]
    if not force and not _newer(HOST_LIB, _sources() + [LIB]):
        return HOST_LIB
    _run([HIPCC, "-x", "c++", *HOST_FLAGS, "-shared", *srcs, "-L" + PKG, "-lrt_gfx950", "-lz", "-Wl,-rpath,$ORIGIN", "-o", HOST_LIB])
    return HOST_LIB


def build_oracle(force=False):
    """The CPU restatement, twice: the build that defines parity (no contraction, -O2) and the same source at
    -O3 for AVX2 + FMA machines (x86-64-v3), which only the bench's CPU baseline times."""
    deps = [ORACLE_SRC, os.path.join(ROOT, "include", "rt_capi.h")]
    common = ["g++", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-function"]
    if force or _newer(ORACLE_LIB, deps):
        _run(common + ["-O2", "-ffp-contract=off", "-o", ORACLE_LIB, ORACLE_SRC])
    if force or _newer(ORACLE_O3_LIB, deps):
        _run(common + ["-O3", "-march=x86-64-v3", "-ffp-contract=fast", "-o", ORACLE_O3_LIB, ORACLE_SRC])
    return ORACLE_LIB


EXAMPLE_SRC = os.path.join(ROOT, "examples", "headless_tracer.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "headless_tracer")


def build_example(force=False):
    """The standalone C++ driver: host classes + C-ABI, linked against the library."""
    if force or _newer(EXAMPLE_BIN, [EXAMPLE_SRC, LIB] + [f for f in _sources() if f.endswith(".h")]):
        _run([HIPCC, "-x", "c++", *HOST_FLAGS, EXAMPLE_SRC, "-L" + PKG, "-lrt_gfx950", "-lz", "-Wl,-rpath,$ORIGIN/../cudaraytracer_amd",
              "-o", EXAMPLE_BIN])
    return EXAMPLE_BIN


def build_all(force=False):
    build_library(force)
    build_host_library(force)
    build_oracle(force)
    build_example(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
