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


@pytest.mark.parametrize("phase,derive_view", [(0, True), (0, False), (1, True), (2, False)])
def test_phase_renders_like_the_oracle(oracle, tmp_path, phase, derive_view):
    exp = rt.Experiment(experiment_file(tmp_path))
    assert exp.errors == []
    scene = exp.scene(phase, derive_view=derive_view)
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
