#!/bin/bash
# Runs on the GPU box: builds tools/pmc_calibrate (if missing) and collects FETCH_SIZE / WRITE_SIZE / raw TCC counters
# for its kernels in separate rocprofv3 passes (counters only + kernel trace).  Usage: tools/pmc_calibrate.sh <outdir under gpurun_out>
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/${1:-pmc_cal}
mkdir -p "$OUT"
[ -x "$REPO/tools/pmc_calibrate" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$REPO/tools/pmc_calibrate" "$REPO/tools/pmc_calibrate.hip"
cd /tmp && export TMPDIR=/tmp
"$REPO/tools/pmc_calibrate" > "$OUT/expected.json"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pass$i" -o pmc -- "$REPO/tools/pmc_calibrate" > /dev/null 2> "$OUT/pass$i.err"
done
du -sh "$OUT"
