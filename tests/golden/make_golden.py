#!/usr/bin/env python3
"""Writes the golden frames and probe tables under tests/golden/.

The reference cannot be built in this environment (its trace path needs the
CUDA toolkit headers and thrust), so these vectors come from the CPU oracle,
which itself is pinned to the reference's unit-test assertions.  They freeze
the oracle + scene generators: a later change to either that alters a hit key
or a radiance bit shows up against these files, and the GPU tests can compare
against them without running the oracle.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cudaraytracer_amd as rt  # noqa: E402
from oracle import pyoracle  # noqa: E402

# name, w, h, size, lights, depth, tight
FRAMES = {
    "cubes_ref": ("cubes", 96, 64, 8, 4, 5, False),
    "mixed_ref": ("mixed", 96, 64, 0, 3, 4, False),
    "soup_tight": ("soup", 96, 64, 5000, 2, 0, True),
    "heightfield_tight": ("heightfield", 96, 64, 5000, 2, 0, True),
}


def frames():
    for key, (name, w, h, size, lights, depth, tight) in FRAMES.items():
        s = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=tight)
        rgba, keys, cnt = pyoracle.render(s.desc, s.opts, s.camera, w, h, threads=8, shadow_any_hit=True)
        np.savez_compressed(os.path.join(HERE, f"frame_{key}.npz"), rgba=rgba, keys=keys.astype(np.int32),
                            counters=np.array([cnt[n] for n in rt.RT_COUNTER_NAMES], dtype=np.int64))
        print(key, {k: v for k, v in cnt.items()})


def probe_inputs():
    rng = np.random.default_rng(20240611)
    n = 256
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    mn = rng.uniform(-4, 2, (n, 3)).astype(np.float32)
    mx = (mn + rng.uniform(0, 3, (n, 3))).astype(np.float32)
    o = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = ((mn + mx) / 2 + rng.normal(scale=1.2, size=(n, 3)) - o).astype(np.float32)  # aimed near the boxes: about half hit
    d[::17, 0] = 0  # axis-parallel rays: infinite reciprocals
    d[::23, 1] = -0.0
    v = rng.uniform(-2, 2, (n, 3, 3)).astype(np.float32)
    to = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    target = (v.mean(axis=1) + rng.normal(scale=0.6, size=(n, 3))).astype(np.float32)
    c = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    return dict(box_min=mn, box_max=mx, box_o=o, box_d=unit(d), box_tmax=rng.uniform(2, 20, n).astype(np.float32),
                tri_v=v, tri_o=to, tri_d=unit(target - to),
                sph_c=c, sph_r=rng.uniform(0.3, 2, n).astype(np.float32), sph_d=unit(c + rng.normal(scale=0.8, size=(n, 3)) - to),
                fr_i=unit(rng.normal(size=(n, 3))), fr_n=unit(rng.normal(size=(n, 3))), fr_ior=rng.uniform(1.0, 2.5, n).astype(np.float32))


def eval_probes(p):
    """Runs the oracle's single-function entry points over the probe inputs."""
    lib = pyoracle.load()
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    big = float(np.finfo(np.float32).max)
    n = len(p["box_min"])
    box = np.array([lib.oracle_box_intersect(f3(p["box_min"][i]), f3(p["box_max"][i]), f3(p["box_o"][i]), f3(p["box_d"][i]),
                                             float(p["box_tmax"][i])) for i in range(n)], dtype=np.int8)
    tri = np.zeros((n, 4), dtype=np.float32)
    sph = np.zeros((n, 2), dtype=np.float32)
    kr = np.zeros(n, dtype=np.float32)
    refr = np.zeros((n, 3), dtype=np.float32)
    for i in range(n):
        out = (C.c_float * 3)()
        hit = lib.oracle_triangle_intersect(f3(p["tri_v"][i, 0]), f3(p["tri_v"][i, 1]), f3(p["tri_v"][i, 2]), f3(p["tri_o"][i]),
                                            f3(p["tri_d"][i]), big, out)
        tri[i] = (hit, out[0], out[1], out[2]) if hit else (0, 0, 0, 0)
        hit = lib.oracle_sphere_intersect(f3(p["sph_c"][i]), float(p["sph_r"][i]), f3(p["tri_o"][i]), f3(p["sph_d"][i]), big, out)
        sph[i] = (hit, out[0]) if hit else (0, 0)
        kr[i] = lib.oracle_fresnel(f3(p["fr_i"][i]), f3(p["fr_n"][i]), float(p["fr_ior"][i]))
        lib.oracle_refract(f3(p["fr_i"][i]), f3(p["fr_n"][i]), float(p["fr_ior"][i]), out)
        refr[i] = list(out)
    return dict(box_hit=box, tri_out=tri, sph_out=sph, fr_kr=kr, fr_refr=refr)


def probes():
    p = probe_inputs()
    r = eval_probes(p)
    np.savez_compressed(os.path.join(HERE, "probes.npz"), **p, **r)
    print("probes: box hits", int(r["box_hit"].sum()), "tri hits", int(r["tri_out"][:, 0].sum()), "sphere hits", int(r["sph_out"][:, 0].sum()),
          "total internal reflection", int((r["fr_refr"] == 0).all(1).sum()))


if __name__ == "__main__":
    frames()
    probes()
