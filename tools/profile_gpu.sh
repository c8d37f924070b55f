#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <config> <outdir under gpurun_out> [GSPLAT_ROUNDS value for the one-at-a-time and PMC runs]
# (a context chooses between one-round and two-round frames from frame times it measures itself; under the profiler's
# per-kernel overhead those times are not the real ones, so the profiled runs pin the setting the context settles on
# in a plain run — bench.py prints it as pairs_round — and skip the settle phase)
set -u
CFG=${1:-c3}
OUT=$(pwd)/gpurun_out/${2:-prof}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
if [ -n "${3:-}" ]; then export GSPLAT_ROUNDS=$3; fi
BENCH="python $REPO/bench.py --config $CFG --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --frames-in-flight 1 --no-host-copy-legs"
if [ -z "${PROFILE_ONLY_PMC:-}" ]; then
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
fi
PMC="python $REPO/bench.py --config $CFG --steps 3 --warmup 2 --settle 0 --no-cpu-baseline --frames-in-flight 1 --no-host-copy-legs"
# the default command (2 frames in flight), for the record: kernel durations there include overlap with the other frame
[ -z "${PROFILE_ONLY_PMC:-}" ] && (unset GSPLAT_ROUNDS; timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace_default" -o trace -- python $REPO/bench.py --config $CFG --no-cpu-baseline --no-host-copy-legs > "$OUT/bench_trace_default.json" 2> "$OUT/trace_default.err")
# counter passes: rocprofv3's counter tool has crashed inside a kernel dispatch on this pool more than once (round 5: the
# WRITE_SIZE passes of c3r and c3d came back empty and were summarised as zeros; round 6: SIGSEGV in both passes of c3r,
# profiles/r06_c3r_pmc_first_try.err) — a pass is repeated, up to three times, until its database exists
pmc_pass() {  # <name> <counter>
  for try in 1 2 3; do
    rm -rf "$OUT/pmc_$1"
    # try 2: one dispatch in flight at a time (the tool's crash sits inside a kernel dispatch); try 3: counters only for the
    # kernels the summaries are made of
    EXTRA=""; SER=0
    if [ $try = 2 ]; then SER=3; fi
    if [ $try = 3 ]; then SER=3; EXTRA="--kernel-include-regex render_kernel|downsweep|project_kernel|emit_kernel|upsweep|boundaries"; fi
    AMD_SERIALIZE_KERNEL=$SER timeout 600 rocprofv3 --pmc $2 --kernel-trace $EXTRA -d "$OUT/pmc_$1" -o $1 -- $PMC > /dev/null 2> "$OUT/pmc_$1.err"
    if [ -f "$OUT/pmc_$1/$1_results.db" ]; then echo "pmc_$1: try $try ok" >> "$OUT/pmc_tries.txt"; return 0; fi
    cp "$OUT/pmc_$1.err" "$OUT/pmc_$1_try$try.err"; echo "pmc_$1: try $try FAILED" >> "$OUT/pmc_tries.txt"
  done
  return 1
}
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
