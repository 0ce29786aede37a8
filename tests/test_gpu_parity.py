"""GPU parity: the HIP render path (through the C-ABI) against the CPU oracle
on the same seeded scenes."""
import numpy as np
import pytest

import cudaraytracer_amd as rt
from parity import assert_parity

pytestmark = pytest.mark.gpu

CASES = [
    # name, w, h, size, lights, depth, tight
    ("cubes", 160, 120, 16, 4, 5, False),
    ("cubes", 160, 120, 16, 4, 5, True),
    ("soup", 160, 96, 20000, 1, 0, False),
    ("soup", 192, 108, 200000, 1, 0, True),
    ("soup", 160, 96, 50000, 10, 0, True),
    ("heightfield", 160, 96, 20000, 2, 0, True),
    ("mixed", 160, 120, 0, 16, 4, False),
    ("mixed", 160, 120, 0, 4, 5, True),
    ("mixed", 161, 67, 0, 2, 0, True),  # ragged tiles, depth 0 with mirror / glass present
]


@pytest.mark.parametrize("name,w,h,size,lights,depth,tight", CASES)
def test_frame_matches_oracle(gpu_ctx, oracle, name, w, h, size, lights, depth, tight):
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=tight)
    ref_rgba, ref_keys, ref_cnt = oracle.render(scene.desc, scene.opts, scene.camera, w, h, threads=8, shadow_any_hit=True)
    rgba, keys, cnt = gpu_ctx.render_frame(scene, want_keys=True, want_counters=True)
    r = assert_parity(rgba, keys, ref_rgba, ref_keys)
    print(name, r)
    # the counter build walks exactly the oracle's traversal: integer-exact test counts
    assert cnt == ref_cnt, (cnt, ref_cnt)
    # and the plain build renders the same frame as the counter build
    rgba2, keys2, _ = gpu_ctx.render_frame(scene, want_keys=True)
    assert np.array_equal(keys, keys2)
    assert np.array_equal(rgba, rgba2, equal_nan=True)
