#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE: one pass each) of the tools/phase_profile.py workload, per kernel and launch;
# usage: pmc_traffic.sh TAG [frames [frame_bytes kind level]]
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
TAG=${1:-t}; N=${2:-8192}; shift; shift
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $C -d $OUT/${TAG}_$C -o pmc --output-format csv -- python3 $ROOT/tools/phase_profile.py $N "$@" > $OUT/${TAG}_$C.log 2>&1 || { tail -5 $OUT/${TAG}_$C.log; exit 1; }
done
python3 - $OUT/${TAG} <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(sys.argv[1] + "_" + c + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("zsa_k_") and int(r["Grid_Size"]) >= 1024 * 64: acc[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, d), cs in sorted(acc.items()):
    print(k, d, {c: "%.2f GB (raw counter x 1 KB)" % (v * 1024 / 1e9) for c, v in cs.items()})
PY
