#!/usr/bin/env python3
"""Summarise tools/prof.sh output: per-wave counter means for the decode kernel."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "vit_pk"
tot = {}
for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    fs = glob.glob("%s/%s/*/*_counter_collection.csv" % (d, sub))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        tot[k] = sum(v) / len(v)
W = tot.get("SQ_WAVES", 1)
for k, v in sorted(tot.items()):
    print("%-24s %.4g  per-wave %.1f" % (k, v, v / W))
for f in glob.glob("%s/kt/*/*_kernel_stats.csv" % d):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print("kernel avg ns", r["AverageNs"], "calls", r["Calls"])
