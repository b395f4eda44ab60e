#!/usr/bin/env python3
"""Prints per-launch and per-frame averages of the SQ counter passes under gpurun_out/prof/<tag>_sq*."""
import csv, collections, glob, sys
tag = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
for d in (tag + "_sq", tag + "_sq2"):
    fs = glob.glob("gpurun_out/prof/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, cs in per.items():
        if not k.startswith("zsa_k_") or "link" in k: continue
        print(k, len(n[k]), "launches")
        for c, v in sorted(cs.items()): print("   %-24s %.4g per launch, %.4g per frame" % (c, v / len(n[k]), v / len(n[k]) / frames))
