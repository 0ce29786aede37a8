#!/usr/bin/env python3
"""Prints per-kernel means of a profiles/run_pmc.sh output directory."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
for f in glob.glob(f"{out}/trace/**/*_kernel_stats.csv", recursive=True):
    print("# kernel stats (ns)")
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us  {r["Percentage"]}%')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "Kernel" not in k or "at::" in k:
            continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("#", k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
