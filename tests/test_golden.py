"""The CPU oracle against the committed golden vectors (tests/golden/, written by
tests/golden/make_golden.py).  The GPU counterpart lives in test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

import cudaraytracer_amd as rt

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402


@pytest.mark.parametrize("key", sorted(make_golden.FRAMES))
def test_oracle_reproduces_golden_frame(oracle, key):
    name, w, h, size, lights, depth, tight = make_golden.FRAMES[key]
    g = np.load(os.path.join(GOLDEN, f"frame_{key}.npz"))
    s = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=tight)
    # thread count and row interleaving must not matter
    rgba, keys, cnt = oracle.render(s.desc, s.opts, s.camera, w, h, threads=3, shadow_any_hit=True)
    assert np.array_equal(keys, g["keys"])
    assert np.array_equal(rgba, g["rgba"], equal_nan=True)
    assert [cnt[n] for n in rt.RT_COUNTER_NAMES] == g["counters"].tolist()


def test_oracle_reproduces_golden_probes():
    g = np.load(os.path.join(GOLDEN, "probes.npz"))
    got = make_golden.eval_probes(g)
    for k, v in got.items():
        assert np.array_equal(v, g[k], equal_nan=True), k
    # the table exercises both outcomes of every predicate
    assert 0 < g["box_hit"].sum() < len(g["box_hit"])
    assert 0 < g["tri_out"][:, 0].sum() < len(g["tri_out"])
    assert 0 < g["sph_out"][:, 0].sum() < len(g["sph_out"])
    tir = (g["fr_refr"] == 0).all(1)
    assert 0 < tir.sum() < len(tir)
    assert np.all(g["fr_kr"][tir] == 1.0)  # total internal reflection <=> kr == 1 (cudaTracer/tracer.cu fresnel)


def test_golden_frames_have_content():
    for key in make_golden.FRAMES:
        g = np.load(os.path.join(GOLDEN, f"frame_{key}.npz"))
        hit = g["keys"][..., 0] >= 0
        assert 0.005 < hit.mean() < 1.0, key
