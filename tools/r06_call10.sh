#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stand_in_rccl" > $O/parity.txt 2>&1; echo "parity rc=$?"; tail -5 $O/parity.txt
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
for B in 1 4; do
GPU_MAX_HW_QUEUES=8 GSPLAT_ROUNDS=off timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$O/trace_b$B -o t -- python $REPO/tools/stripe_batch.py c3 --axis rows --batch $B --in-flight 1 --ranks middle --reps 400 > $REPO/$O/trace_b$B.txt 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$REPO/$O/trace_b$B/**/t_results.db", recursive=True)[0])
rows = db.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by sum(duration) desc").fetchall()
print("B=$B")
for n,c,a,t in rows[:24]:
    import re
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", n)
    print(f"  {(m.group(1) if m else n[:50]):60s} calls {c:6d} avg {a/1e3:8.2f} us total {t/1e6:8.2f} ms")
PY
rm -rf $REPO/$O/trace_b$B
done
