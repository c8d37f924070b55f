#!/bin/bash
# Runs on the GPU box: rocprofv3 --pmc passes (counters only + kernel trace) of a short bench.py run.
# Usage: tools/pmc_gpu.sh <config> <outdir under gpurun_out> "<counters pass 1>" ["<counters pass 2>" ...]
set -u
CFG=$1; OUT=$(pwd)/gpurun_out/$2; shift 2
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline"
i=0
for C in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pass$i" -o pmc -- $CMD > /dev/null 2> "$OUT/pass$i.err"
  tail -2 "$OUT/pass$i.err"
done
if [ ! -f "$REPO/gpurun_out/counters_list.txt" ]; then rocprofv3 -L > "$REPO/gpurun_out/counters_list.txt" 2>&1; fi
du -sh "$OUT"
