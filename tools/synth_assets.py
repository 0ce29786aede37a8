"""Writes a synthetic asset directory with the layout and file formats of the reference's `assets/` folder:

  models/metalBarrel/metal_barrel.obj + .mtl + three 1024x1024 PNG maps   (an OBJ of 244 quads = 488 triangles)
  textures/pattern_106/{diffus,normal,specular}.png                       (the ground plane's maps)

The reference's own files stay where they are (they cannot travel to a GPU box); these stand-ins have the same
shape - exporter-style OBJ text with two spaces after `v`, quads, trailing blanks, 1-based v/vt/vn triples, an .mtl
naming map_Kd / map_Ks / map_bump, RGB and RGBA PNGs - so that the OBJ reader, the PNG decoder, tangent generation
and CLAMP / WRAP texturing run on the GPU path with inputs like the real ones.  Everything is seeded.
"""
import math
import os

import numpy as np
from PIL import Image

SEGMENTS, BANDS = 16, 15


def _noise(rng, n, octaves=5):
    """Smooth value noise in [0, 1], n x n."""
    out = np.zeros((n, n), dtype=np.float32)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        cells = 4 << o
        g = rng.random((cells + 1, cells + 1), dtype=np.float32)
        x = np.linspace(0, cells, n, endpoint=False, dtype=np.float32)
        i = x.astype(np.int32)
        f = x - i
        f = f * f * (3 - 2 * f)
        rows = g[i][:, i] * (1 - f)[None, :] + g[i][:, i + 1] * f[None, :]
        rows2 = g[i + 1][:, i] * (1 - f)[None, :] + g[i + 1][:, i + 1] * f[None, :]
        out += amp * (rows * (1 - f)[:, None] + rows2 * f[:, None])
        total += amp
        amp *= 0.5
    return out / total


def _save(path, rgb, mode="RGB"):
    img = np.clip(rgb * 255.0, 0, 255).astype(np.uint8)
    if mode == "RGBA":
        img = np.concatenate([img, np.full(img.shape[:2] + (1,), 255, dtype=np.uint8)], axis=2)
    Image.fromarray(img, mode).save(path)


def _normal_map(height, strength):
    gy, gx = np.gradient(height)
    n = np.stack([-gx * strength, -gy * strength, np.ones_like(height)], axis=2)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    return n * 0.5 + 0.5


def write_barrel(folder, size=1024, seed=7):
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(seed)
    # profile: radius over height, with two hoops like a steel drum
    ys = np.linspace(0.0, 90.0, BANDS + 1)
    rad = 30.0 + 1.5 * np.exp(-((ys - 30.0) / 3.0) ** 2) + 1.5 * np.exp(-((ys - 60.0) / 3.0) ** 2)
    lines = ["# synthetic stand-in for an exporter-written barrel", "", "mtllib metal_barrel.mtl", ""]
    for b in range(BANDS + 1):
        for s in range(SEGMENTS):
            a = 2 * math.pi * s / SEGMENTS
            lines.append(f"v  {rad[b] * math.cos(a):.4f} {ys[b]:.4f} {-rad[b] * math.sin(a):.4f}")
    base_caps = (BANDS + 1) * SEGMENTS
    for y in (0.0, 90.0):                       # lids: a square inscribed in the rim, split into two quads
        for a in (0.25, 0.75, 1.25, 1.75):
            lines.append(f"v  {21.0 * math.cos(a * math.pi):.4f} {y:.4f} {-21.0 * math.sin(a * math.pi):.4f}")
        for a in (0.5, 1.5):
            lines.append(f"v  {14.85 * math.cos(a * math.pi):.4f} {y:.4f} {-14.85 * math.sin(a * math.pi):.4f}")
    lines.append(f"# {base_caps + 12} vertices")
    lines.append("")
    for b in range(BANDS + 1):
        for s in range(SEGMENTS + 1):
            lines.append(f"vt {s / SEGMENTS:.4f} {b / BANDS:.4f} 0.0000")
    lines += ["vt 0.2500 0.2500 0.0000", "vt 0.7500 0.2500 0.0000", "vt 0.7500 0.7500 0.0000", "vt 0.2500 0.7500 0.0000",
              "vt 0.5000 0.2500 0.0000", "vt 0.5000 0.7500 0.0000", ""]
    for s in range(SEGMENTS):
        a = 2 * math.pi * s / SEGMENTS
        lines.append(f"vn {math.cos(a):.4f} 0.0000 {-math.sin(a):.4f}")
    lines += ["vn 0.0000 -1.0000 0.0000", "vn 0.0000 1.0000 0.0000", "", "g metal_barrel", "usemtl ___default___", "s 1"]
    vt_row = SEGMENTS + 1
    for b in range(BANDS):
        for s in range(SEGMENTS):
            s1 = (s + 1) % SEGMENTS
            v = [b * SEGMENTS + s, b * SEGMENTS + s1, (b + 1) * SEGMENTS + s1, (b + 1) * SEGMENTS + s]
            t = [b * vt_row + s, b * vt_row + s + 1, (b + 1) * vt_row + s + 1, (b + 1) * vt_row + s]
            n = [s, s1, s1, s]
            lines.append("f " + " ".join(f"{v[k] + 1}/{t[k] + 1}/{n[k] + 1}" for k in range(4)) + " ")
    vt_caps = (BANDS + 1) * vt_row
    for lid, (normal, order) in enumerate(((SEGMENTS + 1, (0, 3, 2, 1)), (SEGMENTS + 2, (0, 1, 2, 3)))):
        c = base_caps + 6 * lid
        quads = [(c + 0, c + 4, c + 5, c + 3), (c + 4, c + 1, c + 2, c + 5)]
        tq = [(vt_caps + 0, vt_caps + 4, vt_caps + 5, vt_caps + 3), (vt_caps + 4, vt_caps + 1, vt_caps + 2, vt_caps + 5)]
        for q, t in zip(quads, tq):
            lines.append("f " + " ".join(f"{q[k] + 1}/{t[k] + 1}/{normal}" for k in order) + " ")
    lines.append(f"# {BANDS * SEGMENTS + 4} polygons")
    with open(os.path.join(folder, "metal_barrel.obj"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(folder, "metal_barrel.mtl"), "w") as f:
        f.write("# synthetic\n\nnewmtl ___default___\nKa 0.878 0.878 0.878\nKd 0.878 0.878 0.878\nKs 1 1 1\nd 1.0\nNs 752.6\n\n"
                "map_Kd diffus_stainless_steel.png\nmap_Ks specular_stainless_steel.png\nmap_bump normal_hard_bumps.png\n")
    brushed = _noise(rng, size)
    streaks = np.repeat(rng.random((size, 1), dtype=np.float32), size, axis=1)
    steel = 0.55 + 0.25 * brushed + 0.15 * streaks
    _save(os.path.join(folder, "diffus_stainless_steel.png"), np.stack([steel, steel * 0.98, steel * 0.95], axis=2))
    spec = 0.4 + 0.6 * _noise(rng, size, 3)
    _save(os.path.join(folder, "specular_stainless_steel.png"), np.stack([spec] * 3, axis=2))
    bumps = (np.sin(np.linspace(0, 40 * math.pi, size, dtype=np.float32))[None, :] *
             np.sin(np.linspace(0, 40 * math.pi, size, dtype=np.float32))[:, None] > 0.6).astype(np.float32)
    _save(os.path.join(folder, "normal_hard_bumps.png"), _normal_map(bumps + 0.1 * brushed, 6.0))


def write_plane_textures(folder, size=1024, seed=11):
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(seed)
    n = _noise(rng, size)
    u = np.linspace(0, 8, size, endpoint=False, dtype=np.float32)
    tiles = ((np.floor(u)[None, :] + np.floor(u)[:, None]) % 2).astype(np.float32)
    grout = ((u % 1.0 < 0.04)[None, :] | (u % 1.0 < 0.04)[:, None]).astype(np.float32)
    base = (0.35 + 0.3 * tiles + 0.2 * n) * (1 - 0.6 * grout)
    _save(os.path.join(folder, "diffus.png"), np.stack([base, base * 0.9, base * 0.8], axis=2))
    _save(os.path.join(folder, "normal.png"), _normal_map(0.2 * n - grout, 8.0), mode="RGBA")
    spec = 0.2 + 0.6 * tiles * (1 - grout)
    _save(os.path.join(folder, "specular.png"), np.stack([spec] * 3, axis=2))


def write_assets(root, size=1024):
    """Creates <root>/models/metalBarrel and <root>/textures/pattern_106; returns root."""
    write_barrel(os.path.join(root, "models", "metalBarrel"), size)
    write_plane_textures(os.path.join(root, "textures", "pattern_106"), size)
    return str(root)
