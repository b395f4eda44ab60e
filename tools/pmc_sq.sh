#!/bin/bash
# One SQ counter pass over bench.py (instruction mix); prints per-frame averages for the two codec kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY -d $OUT/${TAG}_sq -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs ${BENCH_ARGS:-} > $OUT/${TAG}_sq.log 2>&1 || { tail -5 $OUT/${TAG}_sq.log; exit 1; }
python3 - $OUT/${TAG}_sq <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if k.startswith("zsa_k_") and "link" not in k: acc[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, _), cs in acc.items():
    for c, v in cs.items(): per[k][c].append(v)
for k, cs in per.items():
    print(k, {c: round(sum(v) / len(v) / 8192 / 1e3, 1) for c, v in cs.items()}, "(thousands per frame)")
PY
