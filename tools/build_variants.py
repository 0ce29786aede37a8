#!/usr/bin/env python3
"""Builds tuning variants of the library into gpurun_out/variants/ (development aid).
usage: tools/build_variants.py name=-DX=1,-DY=2 name2=...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracer_amd import build as b  # noqa: E402

outdir = os.path.join(b.ROOT, "variants")
os.makedirs(outdir, exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    b.build_library(force=True, extra_device_flags=[f for f in flags.split(",") if f], out=os.path.join(outdir, f"librt_{name}.so"), tag="_" + name)
