"""bench.py as the driver runs it (a child process, one GPU), on a small frame so that it takes seconds: the one JSON
line must carry the agreed keys and its figures must follow from one another."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--size", "100000", "--width", "640",
           "--height", "360", "--no-others", *extra]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    return json.loads(lines[0])


def test_bench_line_is_consistent():
    d = run_bench("--cpu-seconds", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays of one frame / time of one step
    rays = d["config"]["rays_per_frame"]
    assert rays == d["config"]["primary"] + d["config"]["shadow"]
    assert d["value"] == pytest.approx(rays / d["ms_per_step"] / 1e3, rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "valu_fp32" and r["unit"] == "TFLOP/s"
    assert 0.0 < r["frac"] <= 1.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(r["algorithmic_flops_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12, rel=1e-6)
    assert 0.0 < r["hbm"]["frac"] <= 1.0
    lanes = r["lane_use"]["trace_closest"]
    assert 0.0 < lanes["lanes_per_node_step"] <= 64.0 and 0.0 < lanes["rays_per_triangle_phase_of_16"] <= 16.0
    # whole-frame algorithmic rate by the wall clock cannot exceed the peak either
    assert r["other_kernels"]["whole frame, by wall clock, frames in flight"]["achieved"] <= r["peak"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_bench_headline_runs_on_either_tree_and_in_order():
    d = run_bench("--no-cpu", "--bvh", "median", "--frames-in-flight", "1")
    assert d["cpu_baseline"] is None
    assert d["config"]["frames_in_flight"] == 1
    assert "median" in d["config"]["workload"]
    assert d["value"] > 0
