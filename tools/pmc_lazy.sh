#!/bin/bash
# SQ issue/wait counters of zsa_k_compress_lazy on the level-5 / level-7 runs (one PMC pass, its own run).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT/${TAG}_lazy_sq -o pmc --output-format csv -- python3 $ROOT/tools/config_sweep.py lazy > $OUT/${TAG}_lazy_sq.log 2>&1 || exit 1
echo collected
