"""Experiment description files (cudaTracer/ExperimentReader.cpp:8-158) read by csrc/host/ExperimentReader.h, and the
scene RayTracer::initScene (cudaTracer/RayTracer.cpp:50-101) assembles for a phase.  CPU only.

The reference repository holds no experiment file and no test of its reader, so the expectations below are read off
the reader's source: parity unpinned by fixtures."""
import json
import math

import numpy as np
import pytest
from PIL import Image

import cudaraytracer_amd as rt
from cudaraytracer_amd import experiment as cli

F = np.float32

FULL = {
    "name": "scaling",
    "phases": [
        {"name": "defaults only", "options": {}},
        {"name": "all keys", "options": {
            "lightCnt": 3, "fov": 60.0, "width": 640, "height": 360, "shadowBias": 0.001, "blockSize": 8, "maxDepth": 2,
            "samples": 4, "gammaCorrection": False}},
        {"name": "model", "options": {
            "model": {"folder": "some/dir/", "fname": "thing.obj", "scale": 0.02, "translate": [1.0, -3.0, 5.5]}, "lightCnt": 1}},
    ],
}


def expect_defaults(p):
    d = p["device"]
    assert (d.width, d.height, d.maxDepth, d.samples, d.blockSize) == (1024, 768, 5, 1, 4)
    assert d.shadowBias == F(0.0001) and d.gammaCorrection == 1 and d.useSceneBvh == 1 and d.useMeshBvhs == 1
    assert tuple(d.backgroundColor) == (F(0.2), F(0.7), F(0.8))
    assert d.scale == F(math.tan(F(90 * 0.5) * F(math.pi) / 180)) or abs(d.scale - 1.0) < 2e-7
    assert d.imageAspectRatio == F(1024) / F(768)
    assert p["lightCnt"] == 10 and p["fov"] == 90.0


def test_phases_start_from_the_defaults_and_override_named_keys(tmp_path):
    path = tmp_path / "exp.json"
    path.write_text(json.dumps(FULL, indent=2))
    e = rt.Experiment(path)
    assert e.errors == []
    assert e.name == "scaling" and [p["name"] for p in e.phases] == ["defaults only", "all keys", "model"]
    expect_defaults(e.phases[0])
    assert e.phases[0]["model"]["fname"] == "" and np.array_equal(e.phases[0]["model"]["transform"], np.eye(4, dtype=F))

    p = e.phases[1]
    d = p["device"]
    assert (d.width, d.height, d.maxDepth, d.samples, d.blockSize, d.gammaCorrection) == (640, 360, 2, 4, 8, 0)
    assert d.shadowBias == F(0.001) and p["lightCnt"] == 3 and p["fov"] == 60.0
    # the reader leaves scale and aspect ratio at the defaults' values (nothing re-derives them after main.cpp:59-61)
    assert d.imageAspectRatio == F(1024) / F(768) and abs(d.scale - 1.0) < 2e-7

    # a phase does not inherit from the phase before it
    p = e.phases[2]
    assert p["device"].width == 1024 and p["device"].maxDepth == 5 and p["lightCnt"] == 1
    assert p["model"]["folder"] == "some/dir/" and p["model"]["fname"] == "thing.obj"
    # identity * scale(s) * translate(t), scale(s) writing only 1/s into [3][3] (Mat.cuh:381-386)
    inv = F(1.0) / F(0.02)
    want = np.eye(4, dtype=F)
    want[3] = [inv * F(1.0), inv * F(-3.0), inv * F(5.5), inv]
    assert np.array_equal(p["model"]["transform"], want)


def test_reading_a_text_equals_reading_the_file(tmp_path):
    path = tmp_path / "exp.json"
    path.write_text(json.dumps(FULL))
    a, b = rt.Experiment(path), rt.Experiment(text=json.dumps(FULL))
    assert a.name == b.name and len(a.phases) == len(b.phases)
    for p, q in zip(a.phases, b.phases):
        assert p["name"] == q["name"] and bytes(p["device"]) == bytes(q["device"])
        assert np.array_equal(p["model"]["transform"], q["model"]["transform"])


def test_transformation_matrix_is_row_major_and_scale_translate_replace_it():
    m = [float(i) + 0.5 for i in range(16)]
    e = rt.Experiment(text=json.dumps({"name": "m", "phases": [
        {"name": "matrix", "options": {"model": {"folder": "", "fname": "a.obj", "transformationMatrix": m}}},
        {"name": "both", "options": {"model": {"folder": "", "fname": "a.obj", "transformationMatrix": m, "translate": [1.0, 2.0, 3.0]}}},
        {"name": "short", "options": {"model": {"folder": "", "fname": "a.obj", "transformationMatrix": m[:12]}}},
    ]}))
    assert np.array_equal(e.phases[0]["model"]["transform"], np.array(m, dtype=F).reshape(4, 4))
    want = np.eye(4, dtype=F)
    want[3, :3] = [1, 2, 3]
    assert np.array_equal(e.phases[1]["model"]["transform"], want)
    assert len(e.errors) == 1 and "must be of length 16" in e.errors[0]
    t = e.phases[2]["model"]["transform"]
    assert np.array_equal(t[:3], np.array(m[:12], dtype=F).reshape(3, 4)) and np.array_equal(t[3], [0, 0, 0, 1])


def test_wrong_types_are_reported_in_the_reference_wording_and_reading_carries_on():
    e = rt.Experiment(text=json.dumps({"name": "t", "phases": [{"name": "p", "options": {
        "fov": 60, "width": "wide", "height": 2.5, "gammaCorrection": 1, "samples": True, "maxDepth": 3,
        "model": {"folder": "f/", "fname": "m.obj", "scale": 2, "translate": [1.0, 2.0]}}}]}), )
    texts = [m.split("\n", 1)[1] for m in e.errors]
    assert all(m.startswith("Failed to parse json file: '<string>'.\n") for m in e.errors)
    assert texts == [
        "The translate model.member must be a float!",
        "The translate model->translate must be of length 3!",
        "'fov' must be of type float!",
        "'width' must be of type int!",
        "'height' must be of type int!",
        "'samples' must be of type int!",
        "'gammaCorrection' must be of type bool!",
    ]
    p = e.phases[0]
    d = p["device"]
    assert p["fov"] == 60.0           # an integer literal is not a float for the reader, but its value is still taken
    assert d.width == 1024            # a string is not
    assert d.height == 2              # a number is converted
    assert d.samples == 1 and d.gammaCorrection == 1 and d.maxDepth == 3
    # scale(2) then translate(1, 2, missing -> 0)
    want = np.eye(4, dtype=F)
    want[3] = [0.5, 1.0, 0.0, 0.5]
    assert np.array_equal(p["model"]["transform"], want)


def test_missing_names_and_phase_list():
    e = rt.Experiment(text='{"phases": [{"options": {"maxDepth": 1}}]}')
    assert [m.split("\n", 1)[1] for m in e.errors] == ["The experiment must have a name, of type string!",
                                                      "The phase must have a name, of type string!"]
    assert e.name == "" and len(e.phases) == 1 and e.phases[0]["name"] == "" and e.phases[0]["device"].maxDepth == 1
    e = rt.Experiment(text='{"name": "n"}')
    assert len(e.phases) == 0 and "must have a list of phases" in e.errors[0]
    # a phase without "options" is the defaults
    e = rt.Experiment(text='{"name": "n", "phases": [{"name": "bare"}]}')
    assert e.errors == [] and len(e.phases) == 1
    expect_defaults(e.phases[0])


def test_at_most_five_phases_are_kept():
    phases = [{"name": f"p{i}", "options": {"maxDepth": i}} for i in range(7)]
    e = rt.Experiment(text=json.dumps({"name": "many", "phases": phases}))
    assert len(e.phases) == 5 and [p["device"].maxDepth for p in e.phases] == [0, 1, 2, 3, 4]
    assert len(e.errors) == 1 and "must not be longer than 5" in e.errors[0]


@pytest.mark.parametrize("text", ["", "{", '{"name": "x", "phases": [}', '{"name": "x",}', '{"name": "x"} trailing', '{"a": 01}',
                                  '{"a": "\\q"}', '{"a": "\\ud800"}', '["unterminated', "{'name': 1}", '{"a": 1.}', '{"a": -}',
                                  '{"a": 1e}', "nul", "[" * 100 + "]" * 100])
def test_invalid_json_is_reported_not_fatal(text):
    e = rt.Experiment(text=text)
    assert len(e.phases) == 0 and len(e.errors) == 1 and "Not a JSON document" in e.errors[0] and "offset" in e.errors[0]


def test_missing_file_is_reported(tmp_path):
    e = rt.Experiment(tmp_path / "nope.json")
    assert len(e.phases) == 0 and len(e.errors) == 1 and "nope.json" in e.errors[0] and "cannot be opened" in e.errors[0]


def test_json_details_escapes_unicode_numbers_nesting_and_unknown_keys():
    text = r'''
    { "ignored": [null, true, {"deep": [[[]]], "n": -0.0e+0}], "name": "caf\u00e9 \ud83d\ude00 \"q\" \\ \/ \t",
      "phases": [ {"name": "tab\there", "extra": {}, "options": {
          "shadowBias": 1E-3, "fov": 4.5e1, "width": -0, "maxDepth": 2147483647, "samples": 2147483648,
          "model": {"folder": "C:\\models\\", "fname": "b\u0061rrel.obj", "scale": 1.0}}} ] }'''
    e = rt.Experiment(text=text)
    assert e.name == 'café \U0001F600 "q" \\ / \t'
    p = e.phases[0]
    assert p["name"] == "tab\there" and p["model"]["folder"] == "C:\\models\\" and p["model"]["fname"] == "barrel.obj"
    d = p["device"]
    assert d.shadowBias == F(1e-3) and p["fov"] == 45.0 and d.width == 0 and d.maxDepth == 2147483647
    # 2147483648 does not fit an int: reported, like rapidjson's IsInt
    assert [m.split("\n", 1)[1] for m in e.errors] == ["'samples' must be of type int!"]


# ---- the scene a phase builds -------------------------------------------------------------------------------------

MSVC_RAND = [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358, 26962, 24464, 5705, 28145, 23281, 16827]


def test_light_ring_draws_from_the_windows_runtime_generator():
    """RayTracer::addLights (RayTracer.cpp:85-101) with the well-known output of an unseeded rand() there."""
    e = rt.Experiment(text=json.dumps({"name": "l", "phases": [{"name": "two", "options": {"lightCnt": 2, "width": 64, "height": 48}}]}))
    s = e.scene(0)
    d = s.desc.contents
    assert d.lightCnt == 2
    u = [F(v) / F(32767.0) for v in MSVC_RAND]
    step = F(2.0) * F(3.14159265358979323846) / F(2)
    for i in range(2):
        r = u[7 * i:7 * i + 7]
        light = d.lights[i]
        assert light.power == r[0] * F(25.0)
        ang = F(i) * step
        want = (F(math.sin(ang)) * (r[1] * F(40.0) + F(10.0)), r[2] * F(25.0), F(math.cos(ang)) * (r[3] * F(40.0) + F(10.0)))
        assert np.allclose(tuple(light.position), want, rtol=1e-6, atol=1e-5)
        assert light.position[1] == want[1]
        assert tuple(light.intensity) == (r[4] * F(0.5), r[5] * F(0.5), r[6] * F(0.5))
    s.close()


def test_phase_without_model_builds_the_default_content():
    e = rt.Experiment(text=json.dumps({"name": "d", "phases": [
        {"name": "dflt", "options": {"lightCnt": 0, "width": 320, "height": 200, "fov": 60.0}}]}))
    s = e.scene(0)
    d = s.desc.contents
    assert d.lightCnt == 0
    assert d.shapeCnt == 102 + 1 + 3 and d.meshCnt == 4   # sphere field + 2 balls, plane, three mesh instances
    assert (s.width, s.height) == (320, 200)
    o = s.opts.contents
    assert abs(o.scale - 1.0) < 2e-7 and o.imageAspectRatio == F(1024) / F(768)   # as the reference leaves them
    s.close()
    s = e.scene(0, derive_view=True)
    o = s.opts.contents
    assert o.imageAspectRatio == F(320) / F(200) and abs(o.scale - math.tan(math.radians(30))) < 1e-6
    s.close()


OBJ = """mtllib tri.mtl
v -1 0 0
v 1 0 0
v 0 2 0
v 0 0 1
vt 0 0
vt 1 0
vt 0 1
vn 0 0 -1
usemtl mat
f 1/1/1 3/3/1 2/2/1
f 1/1/1 4/3/1 3/2/1
"""
MTL = "newmtl mat\nKd 1 1 1\nmap_Kd d.png\nmap_bump n.png\nmap_Ks s.png\n"


def write_model(folder):
    (folder / "tri.obj").write_text(OBJ)
    (folder / "tri.mtl").write_text(MTL)
    Image.new("RGBA", (2, 2), (200, 100, 50, 255)).save(folder / "d.png")
    Image.new("RGBA", (1, 1), (128, 128, 255, 255)).save(folder / "n.png")
    Image.new("RGBA", (1, 1), (255, 255, 255, 255)).save(folder / "s.png")


def test_phase_with_model_loads_it_with_the_phase_transform(tmp_path):
    write_model(tmp_path)
    e = rt.Experiment(text=json.dumps({"name": "m", "phases": [{"name": "tri", "options": {
        "model": {"folder": str(tmp_path) + "/", "fname": "tri.obj", "scale": 0.5, "translate": [0.0, -1.0, 4.0]},
        "lightCnt": 1, "width": 64, "height": 64}}]}))
    assert e.errors == []
    s = e.scene(0)
    d = s.desc.contents
    assert d.shapeCnt == 1 and d.meshCnt == 1 and d.lightCnt == 1 and s.triangles == 2
    m = d.meshes[0]
    assert (m.diffuse.width, m.diffuse.height, m.normals.width) == (2, 2, 1)
    # p -> 0.5 p + (0, -1, 4): the shape's box is the transformed model's box
    shape = d.shapes[0]
    assert np.allclose(tuple(shape.bbmin), (-0.5, -1.0, 4.0), atol=1e-6)
    assert np.allclose(tuple(shape.bbmax)[:2], (0.5, 0.0), atol=1e-6) and abs(shape.bbmax[2] - 4.5) < 1e-5
    # transformInverse * transform = identity (up to the common w factor)
    t = e.phases[0]["model"]["transform"].astype(np.float64)
    inv = np.array(m.transformInverse, dtype=np.float64).reshape(4, 4)
    prod = t @ inv
    assert np.allclose(prod / prod[3, 3], np.eye(4), atol=1e-6)
    s.close()


def test_phase_with_unreadable_model_or_bad_size_is_an_error(tmp_path):
    e = rt.Experiment(text=json.dumps({"name": "m", "phases": [
        {"name": "gone", "options": {"model": {"folder": str(tmp_path) + "/", "fname": "missing.obj"}}},
        {"name": "flat", "options": {"width": 0}},
        {"name": "dark", "options": {"lightCnt": -1}}]}))
    for i in range(3):
        with pytest.raises(rt.RtError):
            e.scene(i)
    with pytest.raises(rt.RtError):
        e.scene(3)


def test_cli_reports_reader_errors_and_missing_phases(tmp_path, capsys):
    bad = tmp_path / "bad.json"
    bad.write_text('{"name": "b", "phases": [{"name": "p", "options": {"fov": 60}}]}')
    assert cli.main([str(bad), "--strict"]) == 2
    err = capsys.readouterr().err
    assert "exp name: b" in err and "phase name: p" in err and "'fov' must be of type float!" in err
    assert cli.main([str(tmp_path / "nope.json")]) == 2
    ok = tmp_path / "ok.json"
    ok.write_text('{"name": "o", "phases": [{"name": "p", "options": {}}]}')
    assert cli.main([str(ok), "3", "--out", ""]) == 1          # no such phase: refused before anything touches a GPU
    assert "phase 3 does not exist" in capsys.readouterr().err


# ---- the standalone C++ driver (examples/headless_tracer.cpp) ------------------------------------------------------

def test_cpp_driver_reads_the_experiment_and_fails_loudly_without_a_gpu(tmp_path):
    """Everything up to rt_create runs on the host: arguments, the reader, the scene.  Without a GPU the program says so
    and exits with status 1 (there is no CPU rendering path)."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "headless_tracer")
    if not os.path.exists(exe):
        pytest.skip("examples/headless_tracer not built")
    doc = tmp_path / "exp.json"
    doc.write_text(json.dumps({"name": "cli", "phases": [{"name": "small", "options": {"width": 64, "height": 48, "lightCnt": 2, "fov": 70}}]}))
    r = subprocess.run([exe, str(doc), "4"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "phase 4 does not exist" in r.stderr
    assert "exp name: cli" in r.stdout and "phase name: small" in r.stdout and "'fov' must be of type float!" in r.stderr
    r = subprocess.run([exe, "a", "1", "extra"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 2 and "usage" in r.stderr
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([exe, str(doc), "0", "--out", str(tmp_path / "f.png")], capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 1 and "rt_create" in r.stderr and not (tmp_path / "f.png").exists()
