#!/usr/bin/env python3
"""Prints per-kernel means of a profiles/run_pmc.sh / run_profile.sh output directory.

  summarize_pmc.py <dir> [--traffic-json profiles/traffic.json --workload KEY] [--summary-out FILE]

--summary-out also writes what is printed to FILE, so that ONE run of this script produces both the committed
summary and the table bench.py reads (tests/test_profiles.py compares the two).
With --traffic-json the HBM bytes per launch (and the vector instructions per launch, SQ_INSTS_VALU) of every kernel
that has FETCH_SIZE and WRITE_SIZE samples are merged into that file:
  hbm = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024
(both counters are in KB; on gfx950 FETCH_SIZE tallies 128-byte fabric reads at 64 bytes, so it is
doubled; for this kernel's divergent 64-byte line reads the factor is uncalibrated, which makes the
figure an upper bound).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--traffic-json")
ap.add_argument("--workload", default="default")
ap.add_argument("--summary-out")
args = ap.parse_args()
out = args.out
lines = []
_print = print


def print(*a):  # noqa: A001 - everything printed is also collected for --summary-out
    _print(*a)
    lines.append(" ".join(str(x) for x in a))



def short(name):
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


for f in glob.glob(f"{out}/trace/**/*_kernel_stats.csv", recursive=True):
    print("# kernel stats")
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.2:
            print(f'{short(r["Name"]):40s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:10.1f} us  {r["Percentage"]}%')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "Kernel" not in k or "at::" in k:
            continue
        agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
for k, d in sorted(agg.items()):
    print("#", k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fetch = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024
        write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
        traffic[k] = {"fetch_size_bytes_raw": fetch, "write_size_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write}
        if "SQ_INSTS_VALU" in d:  # wavefront-wide vector instructions per launch (4 SIMD cycles each)
            traffic[k]["valu_insts_per_launch"] = sum(d["SQ_INSTS_VALU"]) / len(d["SQ_INSTS_VALU"])
        if "SQ_THREAD_CYCLES_VALU" in d:  # the same instructions counted per active lane
            traffic[k]["valu_thread_cycles_per_launch"] = sum(d["SQ_THREAD_CYCLES_VALU"]) / len(d["SQ_THREAD_CYCLES_VALU"])
        if "GRBM_GUI_ACTIVE" in d:  # busy cycles of the launch, summed over the 8 XCDs
            traffic[k]["gui_active_cycles_per_launch"] = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"])
        print(f"   => HBM bytes per launch: 2 x {fetch/1e6:.1f} MB + {write/1e6:.1f} MB = {(2*fetch+write)/1e6:.1f} MB")
        # the same figures, unrounded, as they go into the traffic table
        print("   => table: " + json.dumps(traffic[k], sort_keys=True))
if args.traffic_json and traffic:
    table = {}
    if os.path.exists(args.traffic_json):
        table = json.load(open(args.traffic_json))
    table[args.workload] = traffic
    json.dump(table, open(args.traffic_json, "w"), indent=1, sort_keys=True)
    print("wrote", args.traffic_json, "workload", args.workload)
if args.summary_out:
    with open(args.summary_out, "w") as f:
        f.write("\n".join(lines) + "\n")
