#!/usr/bin/env python3
"""Development aid (CPU only): would a tree over SPLIT triangle boxes need fewer tests per ray on the soup?

Every triangle whose box is longer than --tau along its longest axis is cut into slabs along that axis; each piece gets the
bounds of the clipped triangle and refers to the whole triangle (a leaf then lists a triangle once per piece it holds).  A
surface-area tree (full sweep) is built over the pieces in numpy, handed to the CPU oracle as node arrays, and the
oracle's counters give box and triangle tests per ray.  Density as in the benchmark's soup (one triangle per unit volume,
edges U[-0.75, 0.75]^3), smaller cube.

Not a test (pytest does not collect it); it lives here because it uses the oracle as its counter.

usage: python tests/probe_split_boxes.py [--tris 100000] [--tau 0 0.6 0.4]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.setrecursionlimit(10000)

import hand_scenes as hs  # noqa: E402
from oracle import pyoracle  # noqa: E402


def soup(n, seed=1234):
    rng = np.random.default_rng(seed)
    half = 50.0 * (n / 1e6) ** (1 / 3)
    c = rng.uniform(-half, half, (n, 3))
    e1 = rng.uniform(-0.75, 0.75, (n, 3))
    e2 = rng.uniform(-0.75, 0.75, (n, 3))
    return np.stack([c, c + e1, c + e2], axis=1).astype(np.float32), half


def pieces(tri, tau):
    """-> (lo, hi, triangle id) of the pieces; tau <= 0: one piece per triangle."""
    lo, hi = tri.min(1), tri.max(1)
    ids = np.arange(len(tri))
    if tau <= 0:
        return lo, hi, ids
    ext = hi - lo
    axis = ext.argmax(1)
    m = np.maximum(np.ceil(ext[ids, axis] / tau).astype(int), 1)
    out_lo, out_hi, out_id = [lo[m == 1]], [hi[m == 1]], [ids[m == 1]]
    for parts in range(2, m.max() + 1):
        sel = ids[m == parts]
        if len(sel) == 0:
            continue
        t = tri[sel].astype(np.float64)  # [k, 3, 3]
        ax = axis[sel]
        a0 = lo[sel, ax].astype(np.float64)
        step = ext[sel, ax].astype(np.float64) / parts
        coord = np.take_along_axis(t, ax[:, None, None].repeat(3, 1), axis=2)[:, :, 0]  # [k, 3] vertex coordinate along the axis
        for s in range(parts):
            pa, pb = a0 + s * step, a0 + (s + 1) * step
            pts, ok = [], []
            for v in range(3):
                pts.append(t[:, v])
                ok.append((coord[:, v] >= pa) & (coord[:, v] <= pb))
            for (i, j) in ((0, 1), (1, 2), (2, 0)):
                d = coord[:, j] - coord[:, i]
                for plane in (pa, pb):
                    with np.errstate(divide="ignore", invalid="ignore"):
                        w = (plane - coord[:, i]) / d
                    good = np.isfinite(w) & (w >= 0) & (w <= 1)
                    w = np.where(good, w, 0.0)
                    pts.append(t[:, i] + w[:, None] * (t[:, j] - t[:, i]))
                    ok.append(good)
            pts, ok = np.stack(pts, 1), np.stack(ok, 1)  # [k, 9, 3], [k, 9]
            plo = np.where(ok[:, :, None], pts, np.inf).min(1)
            phi = np.where(ok[:, :, None], pts, -np.inf).max(1)
            # outward by an ulp or two, and never outside the triangle's own box
            plo = np.maximum(np.nextafter(plo.astype(np.float32), -np.inf), lo[sel])
            phi = np.minimum(np.nextafter(phi.astype(np.float32), np.inf), hi[sel])
            keep = np.isfinite(plo).all(1)
            out_lo.append(plo[keep])
            out_hi.append(phi[keep])
            out_id.append(sel[keep])
    return np.concatenate(out_lo).astype(np.float32), np.concatenate(out_hi).astype(np.float32), np.concatenate(out_id)


def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def build(lo, hi, max_leaf=8):
    """Full-sweep surface-area tree -> (nodes [(lo, hi, offset, cnt, axis)] in depth-first order, piece order)."""
    cen = (lo + hi) * 0.5
    nodes, order = [], []

    def rec(ids):
        blo, bhi = lo[ids].min(0), hi[ids].max(0)
        me = len(nodes)
        nodes.append(None)
        n = len(ids)
        best = None
        if n > 1:
            parent = area(blo, bhi)
            for ax in range(3):
                srt = ids[np.argsort(cen[ids, ax], kind="stable")]
                l_lo, l_hi = np.minimum.accumulate(lo[srt], 0), np.maximum.accumulate(hi[srt], 0)
                r_lo, r_hi = np.minimum.accumulate(lo[srt][::-1], 0)[::-1], np.maximum.accumulate(hi[srt][::-1], 0)[::-1]
                k = np.arange(1, n)
                cost = area(l_lo[:-1], l_hi[:-1]) * k + area(r_lo[1:], r_hi[1:]) * (n - k)
                i = int(cost.argmin())
                if best is None or cost[i] < best[0]:
                    best = (float(cost[i]), ax, srt, i + 1)
            if n <= max_leaf:
                best = None
        if best is None or n == 1:
            if n > 63:
                raise RuntimeError("leaf too large")
            nodes[me] = (blo, bhi, len(order), n, 0)
            order.extend(ids.tolist())
            return
        _, ax, srt, cut = best
        if n > max_leaf and cut in (0, n):
            cut = n // 2
        rec(srt[:cut])
        second = len(nodes)
        rec(srt[cut:])
        nodes[me] = (blo, bhi, second, 0, ax)

    rec(np.arange(len(lo)))
    return nodes, np.asarray(order)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=100000)
    ap.add_argument("--tau", type=float, nargs="*", default=[0.0, 0.6, 0.4])
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    a = ap.parse_args()
    tri, half = soup(a.tris)
    for tau in a.tau:
        t0 = time.time()
        lo, hi, tid = pieces(tri, tau)
        nodes, order = build(lo, hi)
        t1 = time.time()
        listed = tri[tid[order]]
        mesh = hs.Mesh([hs.tri(*map(tuple, t)) for t in listed.tolist()],
                       nodes=[(l.tolist(), h.tolist(), int(off), int(cnt), int(ax)) for l, h, off, cnt, ax in nodes])
        dist = half * 2.4
        scene = hs.Scene(a.width, a.height, [mesh], lights=[((0, 4 * half, -dist), (0.5, 0.5, 0.5), 25.0)], origin=(0, 0, -dist))
        scene.opts.scale = float(np.tan(np.radians(30)))
        import ctypes as C
        rgba, keys, cnt = pyoracle.render(C.pointer(scene.desc), C.pointer(scene.opts), C.pointer(scene.camera), a.width, a.height, threads=8,
                                          shadow_any_hit=True)
        rays = cnt["primary"] + cnt["shadow"]
        print(f"tau {tau}: {len(tid)} pieces of {a.tris} triangles ({len(tid) / a.tris:.2f}x), {len(nodes)} nodes, build {t1 - t0:.0f} s | "
              f"{cnt['node'] / rays:.1f} box + {cnt['tri'] / rays:.1f} triangle tests per ray ({rays} rays) | hit pixels {(keys[..., 0] >= 0).mean():.3f} | checksum {float(np.nansum(rgba[..., :3])):.6f}", flush=True)


if __name__ == "__main__":
    main()
