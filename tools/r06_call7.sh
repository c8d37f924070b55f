#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call7; mkdir -p $O
for q in default 8 16 2; do for fl in 2 3; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python bench.py --config c3 --no-cpu-baseline --no-host-copy-legs --frames-in-flight $fl --steps 200 > $O/bench_c3_q${q}_f$fl.json 2> $O/err.txt
  python - <<PY
import json
d=json.load(open("$O/bench_c3_q${q}_f$fl.json"))
print("queues $q in flight $fl: value", round(d["value"],1), "sequential", round(d["sequential_fps"],1), "orbit", round(d["orbit"]["fps"],1))
PY
done; done
unset GPU_MAX_HW_QUEUES
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for c in c4 c1; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-host-copy-legs --steps 200 > $O/bench_${c}_q${q}.json 2> $O/err.txt
  python - <<PY
import json
d=json.load(open("$O/bench_${c}_q${q}.json"))
print("$c queues $q: value", round(d["value"],1), "sequential", round(d["sequential_fps"],1))
PY
done; done
