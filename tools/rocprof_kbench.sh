#!/bin/bash
# rocprofv3 kernel trace of tools/kbench.py for the kernels named in $1; per-kernel statistics
# land in gpurun_out/<tag>/ (tag = $2).  Usage: tools/rocprof_kbench.sh confusion,argmax tag [iters]
set -e
ONLY=$1; TAG=$2; ITERS=${3:-30}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kb -- python3 tools/kbench.py --only "$ONLY" --iters "$ITERS" $KBENCH_ARGS > "$OUT/kbench.log" 2>&1
python3 tools/trace_medians.py "$OUT" > "$OUT/medians.txt"
cat "$OUT/medians.txt"
