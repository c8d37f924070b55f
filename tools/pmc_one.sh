#!/bin/bash
# rocprofv3 --pmc <counters> over a short bench run; prints per-kernel averages.  usage: tools/pmc_one.sh <config> "<counters>" [kernel substring]
set -u
CFG=$1; CNT=$2; PAT=${3:-}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_one
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GSPLAT_ROUNDS=${GSPLAT_ROUNDS:-off} timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d "$OUT" -o pmc -- python $REPO/bench.py --config $CFG --steps 3 --warmup 2 --settle 0 --no-cpu-baseline --frames-in-flight 1 --no-host-copy-legs > /dev/null 2> "$OUT/err.txt"
cd $REPO
python - "$PAT" <<PY
import sqlite3, glob, sys
pat = sys.argv[1]
db = sqlite3.connect(glob.glob("gpurun_out/pmc_one/**/pmc_results.db", recursive=True)[0])
rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for k, c, v, n in rows:
    if pat in k: by.setdefault(k[:110], {})[c] = round(v)
for k, d in by.items(): print(k, d)
PY
