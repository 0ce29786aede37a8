"""Soak run (development aid): the frames-in-flight stress test with many seeds and slot counts."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cudaraytracer_amd as rt
import test_gpu_stress as T

ctx = rt.Context(0)
t0 = time.time()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for r in range(rounds):
    seed = 1000 + r
    orig = np.random.default_rng
    np.random.default_rng = lambda s=None, _seed=seed: orig(_seed)
    real_set = ctx.set_frames_in_flight
    nfl = [2, 5, 8, 12, 16][r % 5]
    ctx.set_frames_in_flight = lambda n, _real=real_set, _nfl=nfl: _real(_nfl if n != 1 else 1)
    try:
        T.test_random_sequence_of_frames_in_flight(ctx)
    finally:
        np.random.default_rng = orig
        ctx.set_frames_in_flight = real_set
    print(f"round {r} seed {seed} in flight {nfl}: ok ({time.time() - t0:.1f} s)", flush=True)
print("soak ok")
