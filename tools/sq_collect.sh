#!/bin/bash
# SQ counters of one kernel on the GPU box (two --pmc passes of eight counters; never together with other trace domains):
#   bash tools/sq_collect.sh OUTDIR KERNEL_SUBSTRING -- python3 tools/one_wino.py 64 64 64
# The command after `--` must be the program itself (python3 ..., ./binary): no env / bash -c / taskset / shebang hop --
# rocprofv3's preloaded library has initialised the GPU by then and the box refuses an exec from such a process.
# OUTDIR is scratch: it is created, summarised and removed.
set -eu
if [ "$#" -lt 4 ] || [ "$3" != "--" ]; then
    echo "usage: $0 OUTDIR KERNEL_SUBSTRING -- python3 script.py args..." >&2; exit 2
fi
OUT=$1; K=$2; shift 3
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
case "$OUT" in ""|"/"|".") echo "refusing OUTDIR '$OUT'" >&2; exit 2;; esac
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS"
rm -rf -- "$OUT"; mkdir -p "$OUT"
for pass in 1 2; do
    if [ "$pass" = 1 ]; then P=$P1; else P=$P2; fi
    if ! rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/p$pass" -- "$@" > "$OUT/p$pass.log" 2>&1; then
        echo "rocprofv3 pass $pass failed:" >&2; tail -20 "$OUT/p$pass.log" >&2; exit 1
    fi
done
python3 tools/sq_summary.py "$K" "$OUT/p1" "$OUT/p2"
rm -rf -- "$OUT"
