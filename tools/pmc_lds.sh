#!/bin/bash
# SQ counter passes over bench.py for whichever compress kernel the launch uses (run through gpurun from the repo root)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/${TAG}_sq -o pmc --output-format csv -- $BENCH > $OUT/${TAG}_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVES -d $OUT/${TAG}_sq2 -o pmc --output-format csv -- $BENCH > $OUT/${TAG}_sq2.log 2>&1 || exit 1
echo collected
