"""Soak: the same frame rendered many times with frames in flight; every surface must equal the first one bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPU_MAX_HW_QUEUES"] = "32"
import cudaraytracer_amd as rt

def soak(name, w, h, size, lights, depth, frames, nf, **kw):
    scene = rt.HostScene(name, w, h, size, lights, depth, tight_bvh=True, **kw)
    with rt.Context(0) as ctx:
        ctx.set_frames_in_flight(nf)
        ds = ctx.upload(scene)
        surfs = [ctx.surface(w, h) for _ in range(nf)]
        pitch = surfs[0][1]
        ctx.render(ds, scene.opts, scene.camera, surfs[0][0], pitch)
        ctx.sync()
        ref = ctx.download_surface(surfs[0][0], pitch, w, h).copy()
        bad = 0
        t0 = time.time()
        done = 0
        while done < frames:
            for i in range(nf):
                ctx.render(ds, scene.opts, scene.camera, surfs[i][0], pitch)
            ctx.sync()
            for i in range(nf):
                got = ctx.download_surface(surfs[i][0], pitch, w, h)
                if not np.array_equal(got, ref, equal_nan=True):
                    bad += 1
                    print(name, "frame", done + i, "differs in", int((got != ref).any(-1).sum()), "pixels", flush=True)
            done += nf
            if done % (nf * 20) == 0:
                print(name, done, "frames", f"{time.time()-t0:.0f}s", flush=True)
        print(name, "frames", done, "differing", bad, flush=True)
        return bad

bad = 0
bad += soak("soup", 1920, 1080, 1000000, 1, 0, 1200, 12, sah_bvh=True)
bad += soak("mixed", 640, 480, 0, 8, 4, 1600, 16)
bad += soak("cubes", 320, 240, 16, 4, 5, 3200, 16)
sys.exit(1 if bad else 0)
