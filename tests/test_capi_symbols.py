"""The C-ABI library loads without a GPU and exports every entry point that
include/rt_capi.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import cudaraytracer_amd as rt
from cudaraytracer_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_capi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"static inline[^{;]*\{.*?\n\}", "", text, flags=re.S)  # inline helpers are not exported
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    lib = rt.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in rt_capi.h but not exported"
        assert s in N.C_ABI, f"{s} has no ctypes prototype"
    for s in N.C_ABI:
        assert s in syms, f"{s} bound in Python but not declared in rt_capi.h"


def test_library_exports_only_the_declared_entry_points():
    """The shipped library carries no test scaffolding: its dynamic symbol table is exactly include/rt_capi.h."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_symbols()


def test_struct_sizes_match_reference_layouts():
    # sizes measured from the reference's headers (LP64): Vertex 56, LinearNode 32, Triangle 12,
    # Light 28, Material 32, DeviceOpts 48
    assert C.sizeof(N.RtVertex) == 56
    assert C.sizeof(N.RtLinearNode) == 32
    assert C.sizeof(N.RtTriangle) == 12
    assert C.sizeof(N.RtLight) == 28
    assert C.sizeof(N.RtMaterial) == 32
    assert C.sizeof(N.RtDeviceOpts) == 48
    assert N.RtLinearNode.offset.offset == 24 and N.RtLinearNode.primCnt.offset == 28 and N.RtLinearNode.axis.offset == 30
    assert N.RtVertex.position.offset == 44 and N.RtVertex.texCoord.offset == 36
    o = N.RtDeviceOpts
    assert (o.backgroundColor.offset, o.shadowBias.offset, o.maxDepth.offset, o.gammaCorrection.offset, o.useMeshBvhs.offset) == (16, 28, 36, 44, 46)


def test_rt_create_fails_loudly_without_gpu_or_reports_device():
    import torch

    lib = rt.load()
    if torch.cuda.is_available():
        return  # covered by the gpu tests
    ctx = lib.rt_create(0)
    assert not ctx
    assert b"HIP" in lib.rt_last_error(None) or b"device" in lib.rt_last_error(None)


def test_default_options_match_reference_start_values():
    # cudaTracer/main.cpp:44-64
    s = rt.HostScene("cubes", 1024, 768, 2, 4, 5)
    o = s.opts.contents
    assert (o.width, o.height, o.maxDepth, o.samples, o.blockSize) == (1024, 768, 5, 1, 4)
    assert abs(o.shadowBias - 1e-4) < 1e-10 and o.gammaCorrection == 1 and o.useSceneBvh == 1 and o.useMeshBvhs == 1
    assert [round(v, 3) for v in o.backgroundColor] == [0.2, 0.7, 0.8]
    assert abs(o.scale - 1.0) < 1e-6 and abs(o.imageAspectRatio - 1024 / 768) < 1e-6
