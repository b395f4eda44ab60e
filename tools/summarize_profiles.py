#!/usr/bin/env python3
"""Turns gpurun_out/prof/<tag>_* (tools/collect_profiles.sh) into the committed summaries under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc.json (per-kernel, per-launch counter averages,
L2-miss fabric traffic from FETCH_SIZE / WRITE_SIZE, raw and with the gfx950 x2 read correction of MI355X_MICROARCH.md as an upper bound) and the bench line of the traced run."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")


def find(d, suffix):
    r = glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True)
    return r[0] if r else None


ks = find(tag + "_trace", "kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, tag + "_kernel_stats.csv"))
if os.path.exists(os.path.join(src, tag + "_trace.log")):
    for line in open(os.path.join(src, tag + "_trace.log"), errors="replace"):
        if line.startswith('{"metric"'):
            open(os.path.join(dst, tag + "_bench_under_trace.json"), "w").write(line)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("_fetch", "_write", "_sq", "_sq2"):
    f = find(tag + d, "counter_collection.csv")
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("zsa_k_"):
            continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            pmc[k][c].append(v)
out = {}
for k, cs in pmc.items():
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["launches_averaged"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        # rocprofv3 reports KiB.  Both counters are the L2's memory-side (fabric) request counters: they count L2 misses,
        # Infinity-Cache hits INCLUDED (guide, MI355X_MICROARCH.md:309-321), so this is fabric traffic, an upper bound of what
        # reaches HBM.  On gfx950 FETCH_SIZE tallies a 128-B request of a wide coalesced read at 64 B; scattered 4-byte cell
        # reads are 64-B requests and are counted exactly.  These kernels mix both, so the raw figure (a lower bound of the
        # fabric bytes) and the x2 figure (an upper bound) are both given and nothing is corrected silently.
        out[k]["fabric_read_bytes_raw"] = out[k]["FETCH_SIZE"] * 1024
        out[k]["fabric_read_bytes_x2_upper_bound"] = out[k]["FETCH_SIZE"] * 1024 * 2
        out[k]["fabric_write_bytes"] = out[k]["WRITE_SIZE"] * 1024
        out[k]["fabric_traffic_bytes_raw"] = out[k]["fabric_read_bytes_raw"] + out[k]["fabric_write_bytes"]
        out[k]["fabric_traffic_bytes_x2_upper_bound"] = out[k]["fabric_read_bytes_x2_upper_bound"] + out[k]["fabric_write_bytes"]
# tie the summary to the build and workload it was taken on (bench.py quotes it only when both match)
sys.path.insert(0, ROOT)
import bench
out["build_id"] = bench.build_id()
out["workload"] = {"frames": 8192, "frame_bytes": 131072, "level": 1, "command": "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs"}
json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
