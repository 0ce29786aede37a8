"""Headless experiment runner: `python -m cudaraytracer_amd.experiment FILE [PHASE]`.

The reference runs one phase of an experiment file when it is started with
`<file> <phaseIdx>` on its command line (cudaTracer/main.cpp:97-135): it reads
the file with the defaults of loadDefaults(), builds the scene for that phase's
options, renders `experimentRounds` = 1 frame with dt = 0.016 under a named
timer, dumps the frame as a PNG and logs the timer.  This module does the same
through the C++ driver classes on one MI355X; the timer is printed as one JSON
line instead of being written by the reference's Profiler.
"""
import argparse
import json
import sys
import time

from .api import Experiment, RtError


def run_phase(exp, phase_idx, rounds=1, dt=0.016, png_path=None, asset_dir=None, device=0, tight_bvh=False, device_bvh=False,
              derive_view=False, sah_bvh=False):
    """Builds phase `phase_idx`, renders `rounds` frames with the C++ driver loop and returns (last frame, record)."""
    if not 0 <= phase_idx < len(exp.phases):
        raise RtError(f"experiment '{exp.name}' has {len(exp.phases)} phase(s); phase {phase_idx} does not exist")
    phase = exp.phases[phase_idx]
    t0 = time.perf_counter()
    scene = exp.scene(phase_idx, asset_dir=asset_dir, tight_bvh=tight_bvh, device_bvh=device_bvh, derive_view=derive_view, sah_bvh=sah_bvh)
    t1 = time.perf_counter()
    try:
        rgba = scene.run_frames(rounds, dt, png_path=png_path, device=device)
        t2 = time.perf_counter()
        record = {
            "experiment": exp.name,
            "phase": phase["name"],
            "phase_index": phase_idx,
            "width": scene.width,
            "height": scene.height,
            "triangles": scene.triangles,
            "lights": phase["lightCnt"],
            "maxDepth": phase["device"].maxDepth,
            "samples": phase["device"].samples,
            "rounds": rounds,
            "scene_build_s": round(t1 - t0, 6),
            # upload + rounds x (lights, render, wait) + read-back (+ PNG), as the reference's timer brackets it
            "run_s": round(t2 - t1, 6),
            "png": png_path,
        }
    finally:
        scene.close()
    return rgba, record


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("file", help="experiment description (JSON)")
    ap.add_argument("phase", nargs="?", type=int, default=0, help="phase index (default 0)")
    ap.add_argument("--rounds", type=int, default=1, help="frames to render (the reference renders 1)")
    ap.add_argument("--out", default=None, help="PNG path of the last frame (default: <experiment>_<phase>.png; '' = none)")
    ap.add_argument("--assets", default=None, help="asset directory for phases without a model (default: procedural stand-ins)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--tight-bvh", action="store_true", help="build BVHs from a true empty box")
    ap.add_argument("--sah-bvh", action="store_true", help="cut mesh BVHs by the binned surface-area heuristic (host build) instead of the reference's median cut")
    ap.add_argument("--device-bvh", action="store_true", help="build mesh BVHs on the GPU")
    ap.add_argument("--derive-view", action="store_true", help="re-derive scale / aspect ratio from the phase's fov, width, height")
    ap.add_argument("--strict", action="store_true", help="stop when the reader reported anything")
    args = ap.parse_args(argv)

    exp = Experiment(args.file)
    print(f"file: {args.file}\nexp name: {exp.name}", file=sys.stderr)
    for p in exp.phases:
        print(f"\tphase name: {p['name']}\n\t\tphase model: {p['model']['fname']}", file=sys.stderr)
    for m in exp.errors:
        print(m, file=sys.stderr)
    if exp.errors and (args.strict or not exp.phases):
        return 2
    out = args.out
    if out is None and 0 <= args.phase < len(exp.phases):
        out = f"{exp.name}_{exp.phases[args.phase]['name']}.png".replace("/", "_").replace(" ", "_")
    try:
        _, record = run_phase(exp, args.phase, rounds=args.rounds, png_path=out or None, asset_dir=args.assets, device=args.device,
                              tight_bvh=args.tight_bvh, device_bvh=args.device_bvh, derive_view=args.derive_view, sah_bvh=args.sah_bvh)
    except RtError as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    print(json.dumps(record))
    return 0


if __name__ == "__main__":
    sys.exit(main())
