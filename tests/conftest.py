import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def rtlib():
    import cudaraytracer_amd

    return cudaraytracer_amd.load()


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    import cudaraytracer_amd as rt

    ctx = rt.Context(0)
    # The shared context walks even the small test frames the way large frames are walked (two launches per trace:
    # min / max box test, then the literal one for rays along axis planes); contexts that tests create themselves keep
    # the default, under which small frames take the literal test in one launch.
    ctx.set_box_test_split(0)
    # RT_TEST_UNLIT_SKIP=1 runs the whole GPU suite with rt_set_unlit_shadow_skip on: every comparison with the oracle must hold as it is
    if os.environ.get("RT_TEST_UNLIT_SKIP") == "1":
        ctx.set_unlit_shadow_skip(True)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def synthetic_assets(tmp_path_factory):
    """An asset directory shaped like the reference's (OBJ barrel + 1024x1024 PNG maps + plane textures), written once."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_assets

    return synth_assets.write_assets(tmp_path_factory.mktemp("assets"))
