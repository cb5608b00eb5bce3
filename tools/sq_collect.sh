#!/bin/bash
# SQ counters of one kernel on the GPU box (two --pmc passes of eight counters; never together with other trace domains):
#   bash tools/sq_collect.sh OUTDIR KERNEL_SUBSTRING -- python3 tools/one_wino.py 64 64 64
OUT=$1; K=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS"
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/p1 -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/p2 -- "$@" > /dev/null 2>&1
python3 tools/sq_summary.py "$K" $OUT/p1 $OUT/p2
rm -rf $OUT
