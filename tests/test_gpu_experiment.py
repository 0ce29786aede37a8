"""Experiment files end to end on the GPU: a phase read by the ExperimentReader mirror, its scene assembled like
RayTracer::initScene, one round rendered by the C++ driver loop (cudaTracer/main.cpp:124-132) and dumped; the
oracle renders the same scene state."""
import json

import numpy as np
import pytest
from PIL import Image

import cudaraytracer_amd as rt
from cudaraytracer_amd import experiment as cli
from parity import assert_parity, to_8bit
from test_experiment_reader import write_model

pytestmark = pytest.mark.gpu


def experiment_file(tmp_path):
    write_model(tmp_path)
    doc = {"name": "gpu run", "phases": [
        {"name": "default content", "options": {"lightCnt": 3, "width": 160, "height": 120, "fov": 75.0, "maxDepth": 3}},
        # the one-argument scale puts 1/s into the matrix's w: rays enter the mesh through a projective divide
        {"name": "scaled model", "options": {
            "model": {"folder": str(tmp_path) + "/", "fname": "tri.obj", "scale": 2.5, "translate": [0.5, -2.0, 5.0]},
            "lightCnt": 2, "width": 96, "height": 80, "maxDepth": 1, "gammaCorrection": False}},
        {"name": "four samples", "options": {"lightCnt": 1, "width": 64, "height": 48, "samples": 4, "maxDepth": 2}},
    ]}
    path = tmp_path / "exp.json"
    path.write_text(json.dumps(doc))
    return path


@pytest.mark.parametrize("phase,derive_view,sah", [(0, True, False), (0, False, False), (1, True, False), (2, False, False), (0, True, True),
                                                   (1, True, True)])
def test_phase_renders_like_the_oracle(oracle, tmp_path, phase, derive_view, sah):
    exp = rt.Experiment(experiment_file(tmp_path))
    assert exp.errors == []
    scene = exp.scene(phase, derive_view=derive_view, sah_bvh=sah, tight_bvh=sah)
    w, h = scene.width, scene.height
    png = tmp_path / "frame.png"
    rgba = scene.run_frames(1, 0.016, str(png))   # one round: lights turn by dt, render, read back, dump
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8)
    assert_parity(rgba, None, ref_rgba, None)
    assert (ref_keys[..., 0] >= 0).mean() > 0.05                      # the phase's geometry is in view
    if phase == 1:
        assert ref_cnt["tri"] > 0 and ref_cnt["mesh_entry"] > 0
    dumped = np.asarray(Image.open(png).convert("RGBA")).astype(np.int32)
    assert dumped.shape == (h, w, 4) and np.abs(dumped - to_8bit(ref_rgba)).max() <= 1


def test_command_line_runs_one_phase(tmp_path, capsys):
    path = experiment_file(tmp_path)
    out = tmp_path / "cli.png"
    assert cli.main([str(path), "1", "--out", str(out), "--derive-view", "--device-bvh", "--tight-bvh"]) == 0
    captured = capsys.readouterr()
    record = json.loads(captured.out.strip().splitlines()[-1])
    assert record["experiment"] == "gpu run" and record["phase"] == "scaled model" and record["phase_index"] == 1
    assert (record["width"], record["height"], record["triangles"], record["lights"], record["rounds"]) == (96, 80, 2, 2, 1)
    assert record["run_s"] > 0 and "exp name: gpu run" in captured.err
    assert Image.open(out).size == (96, 80)


GRID_MTL = "newmtl mat\nKd 1 1 1\nmap_Kd d.png\nmap_bump n.png\nmap_Ks s.png\n"


def write_grid_model(folder):
    """Two 8x8-quad grids with vertices on integer coordinates, facing a camera at the origin that looks along +z: box
    planes and triangle edges pass exactly through the camera axis (x = 0, y = 0)."""
    lines = ["mtllib grid.mtl"]
    for z in (5.0, 7.0):
        for y in range(-4, 5):
            for x in range(-4, 5):
                lines.append(f"v {x} {y} {z}")
    for y in range(9):
        for x in range(9):
            lines.append(f"vt {x / 8} {y / 8}")
    lines += ["vn 0 0 -1", "usemtl mat"]
    for layer in range(2):
        base = layer * 81
        for y in range(8):
            for x in range(8):
                a, b, c, d = (base + y * 9 + x + 1, base + y * 9 + x + 2, base + (y + 1) * 9 + x + 2, base + (y + 1) * 9 + x + 1)
                ta, tb, tc, td = y * 9 + x + 1, y * 9 + x + 2, (y + 1) * 9 + x + 2, (y + 1) * 9 + x + 1
                if (x + y + layer) % 3 == 0:
                    continue  # holes, so that rays also reach the second layer
                # counter-clockwise seen from the camera side (-z): the front face
                lines.append(f"f {a}/{ta}/1 {d}/{td}/1 {c}/{tc}/1 {b}/{tb}/1")
    (folder / "grid.obj").write_text("\n".join(lines) + "\n")
    (folder / "grid.mtl").write_text(GRID_MTL)
    Image.new("RGBA", (4, 4), (180, 120, 60, 255)).save(folder / "d.png")
    Image.new("RGBA", (1, 1), (128, 128, 255, 255)).save(folder / "n.png")
    Image.new("RGBA", (1, 1), (200, 200, 200, 255)).save(folder / "s.png")


@pytest.mark.parametrize("w,h", [(33, 33), (5, 5), (1, 1), (7, 3), (64, 48)])
@pytest.mark.parametrize("tight", [False, True])
def test_rays_along_box_planes_and_triangle_edges(gpu_ctx, oracle, tmp_path, w, h, tight):
    """Odd frame sizes put a primary ray exactly on the camera axis: direction (0, 0, 1), so two reciprocals are
    infinite, and with box planes at x = 0 and y = 0 the slab test forms 0 * inf = NaN (BoundingBox.cuh:115-139);
    the ray also runs along shared triangle edges, where u or v is exactly 0 (Mesh.cuh:190-203).  Hit keys, radiance
    and the test counts must equal the oracle's, with the reference's box bounds and with tight ones."""
    write_grid_model(tmp_path)
    exp = rt.Experiment(text=json.dumps({"name": "axes", "phases": [{"name": "grid", "options": {
        "model": {"folder": str(tmp_path) + "/", "fname": "grid.obj"}, "lightCnt": 2, "width": w, "height": h, "fov": 90.0,
        "maxDepth": 0}}]}))
    assert exp.errors == []
    scene = exp.scene(0, derive_view=True, tight_bvh=tight)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=4, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    r = assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert r["key_mismatches"] == 0 and cnt == ref_cnt, (r, cnt, ref_cnt)
    if w >= 33:
        assert (ref_keys[..., 0] >= 0).mean() > 0.3 and ref_cnt["tri"] > 0


def test_texture_coordinates_outside_the_unit_square(gpu_ctx, oracle, tmp_path):
    """A model whose UVs run from -2 to 3 over a 5x3 texture of distinct texels (CLAMP addressing, CudaMesh.h:85-104):
    bilinear taps left of texel 0 and right of the last one, the fp64 weight path (Texture.cuh:33-69), normal-map and
    specular taps included."""
    write_grid_model(tmp_path)
    text = (tmp_path / "grid.obj").read_text().splitlines()
    out = []
    for line in text:
        if line.startswith("vt "):
            _, u, v = line.split()
            out.append(f"vt {float(u) * 5 - 2:.6f} {float(v) * 5 - 2:.6f}")
        else:
            out.append(line)
    (tmp_path / "grid.obj").write_text("\n".join(out) + "\n")
    rng = np.random.default_rng(5)
    for name in ("d.png", "n.png", "s.png"):
        px = rng.integers(0, 256, size=(3, 5, 4), dtype=np.uint8)
        px[..., 3] = 255
        if name == "n.png":
            px[..., 2] = 255   # normals that still point out of the surface
        Image.fromarray(px, "RGBA").save(tmp_path / name)
    exp = rt.Experiment(text=json.dumps({"name": "uv", "phases": [{"name": "grid", "options": {
        "model": {"folder": str(tmp_path) + "/", "fname": "grid.obj"}, "lightCnt": 3, "width": 120, "height": 90, "fov": 90.0,
        "maxDepth": 0}}]}))
    scene = exp.scene(0, derive_view=True, tight_bvh=True)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, 120, 90, threads=4, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    r = assert_parity(rgba, keys, ref_rgba, ref_keys)
    assert r["key_mismatches"] == 0 and cnt == ref_cnt and r["max_rel"] < 1e-5, r
    hit = ref_keys[..., 0] >= 0
    assert hit.mean() > 0.3 and len(np.unique(to_8bit(ref_rgba)[hit][:, :3], axis=0)) > 50   # the texture shows
