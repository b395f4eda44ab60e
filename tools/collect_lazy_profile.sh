#!/bin/bash
# rocprofv3 kernel trace of the level-5 / level-7 runs (tools/config_sweep.py lazy), for DESIGN.md section 4.8.
# Output: gpurun_out/prof/<tag>_lazy_trace; copy the kernel_stats.csv into profiles/ by hand.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_lazy_trace -o trace --output-format csv -- python3 $ROOT/tools/config_sweep.py lazy > $OUT/${TAG}_lazy_trace.log 2>&1 || exit 1
echo collected
