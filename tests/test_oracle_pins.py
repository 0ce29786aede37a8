"""Pins the CPU oracle (and the host API classes) to the reference's own unit-test
assertions: cudaTracer/VectorTest.cpp, Tests.cpp, BoundingBoxTest.cpp, BvhTest.cpp.
Those are the only known answers the reference itself holds for this path; for
traversal order, triangle / sphere tests, Fresnel / refraction, texture filtering and
the tie rule, tests/test_known_answers.py adds answers derived by hand from the
reference's formulas."""
import ctypes as C

import numpy as np


def _run(fn):
    buf = C.create_string_buffer(1 << 16)
    failures = fn(buf, len(buf))
    log = buf.value.decode()
    return failures, log


def test_oracle_reproduces_reference_unit_tests(oracle):
    failures, log = _run(oracle.load().oracle_selftest)
    assert failures == 0, log
    # every reference test file is represented
    for tag in ("VectorTest.cpp", "Tests.cpp", "BoundingBoxTest.cpp", "BvhTest.cpp"):
        assert tag in log
    assert log.count("PASS") >= 35


def test_host_api_reproduces_reference_unit_tests(rtlib):
    failures, log = _run(rtlib.rth_selftest)
    assert failures == 0, log
    assert log.count("PASS") >= 45
    for tag in ("VectorTest.cpp", "Tests.cpp", "BoundingBoxTest.cpp", "BvhTest.cpp"):
        assert tag in log


def test_box_pad_factor_is_one_plus_3_ulp(oracle):
    # BoundingBox.cuh:124: 1 + 2*gamma(3) evaluated in fp32
    f = np.float32(oracle.load().oracle_box_pad_factor())
    assert f.view(np.uint32) == 0x3F800003


def test_slab_test_known_answers(oracle):
    """BoundingBoxTest.cpp:182-240 through the probe entry point, incl. +/-inf reciprocals."""
    lib = oracle.load()
    f3 = lambda *v: (C.c_float * 3)(*v)
    o = f3(0, 0, 0)
    big = np.finfo(np.float32).max
    cases = [
        ((-1, -1, -1), (1, 1, 1), (0, 0, 1), 1),
        ((-1, -1, 1), (1, 1, 3), (0, 0, 1), 1),
        ((0.5, 0.5, 1), (1, 1, 3), (0, 0, 1), 0),
        ((-1, -1, -1), (1, 1, 1), (0, 0, -1), 1),
        ((-1, -1, -1), (1, 1, -3), (0, 0, -1), 1),  # inverted z extents still hit
        ((0.5, 0.5, -1), (1, 1, -3), (0, 0, -1), 0),
    ]
    for mn, mx, d, want in cases:
        assert lib.oracle_box_intersect(f3(*mn), f3(*mx), o, f3(*d), big) == want, (mn, mx, d)
    # the reach of the ray culls a box that starts beyond it
    assert lib.oracle_box_intersect(f3(-1, -1, 5), f3(1, 1, 6), o, f3(0, 0, 1), 4.0) == 0
    assert lib.oracle_box_intersect(f3(-1, -1, 5), f3(1, 1, 6), o, f3(0, 0, 1), 5.5) == 1
