#!/bin/bash
# SQ instruction counters of one compress + decompress launch (tools/phase_profile.py workload); usage: pmc_quick.sh TAG [frames [frame_bytes kind level]]
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
TAG=${1:-q}; N=${2:-8192}; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY -d $OUT/${TAG}_q -o pmc --output-format csv -- python3 $ROOT/tools/phase_profile.py $N "$@" > $OUT/${TAG}_q.log 2>&1 || { tail -5 $OUT/${TAG}_q.log; exit 1; }
python3 - $OUT/${TAG}_q $N <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]; N = int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if k.startswith("zsa_k_") and int(r["Grid_Size"]) >= 1024 * 64: acc[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, d), cs in sorted(acc.items()):
    print(k, d, {c.replace("SQ_", ""): round(v / N / 1e3, 1) for c, v in cs.items()}, "(thousands per frame)")
PY
