#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, then PMC passes (each in a run of its own: FETCH_SIZE, WRITE_SIZE, SQ issue/LDS counters).
# Output: gpurun_out/prof/<tag>_*.  Summaries are copied into profiles/ by tools/summarize_profiles.py.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace --output-format csv -- $BENCH > $OUT/${TAG}_trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o pmc --output-format csv -- $BENCH > $OUT/${TAG}_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write -o pmc --output-format csv -- $BENCH > $OUT/${TAG}_write.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/${TAG}_sq -o pmc --output-format csv -- $BENCH > $OUT/${TAG}_sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVES -d $OUT/${TAG}_sq2 -o pmc --output-format csv -- $BENCH > $OUT/${TAG}_sq2.log 2>&1 || exit 1
echo collected
