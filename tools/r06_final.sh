#!/bin/bash
# round 6, what is left to run once the GPU pool reopens — in order of what the round's claims need:
#   1. the whole -m gpu suite at HEAD (live lists: both forms are parametrised into the block-cull / batch tests)
#   2. live lists A/B on the stripe model (GSPLAT_LIVE_LISTS=off / on, c3 + c4, batches of 1 and 4)
#   3. the round's collection (tools/collect_final.sh r06: counters, bench lines, stripe models)
# usage: tools/r06_final.sh <commit> [skip-collect]
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r06; mkdir -p $O
export GSPLAT_COMMIT=${1:-unknown}
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/rc.txt; tail -4 $O/gpu_suite.txt
cp gpurun_out/twin_report_*.json $O/ 2>/dev/null
export GPU_MAX_HW_QUEUES=8
for ll in off on; do for c in c3 c4; do
  GSPLAT_LIVE_LISTS=$ll GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py $c --axis rows --batch 1,4 --in-flight 1,3,4 --ranks middle,slowest > $O/stripe_batch_${c}_live_$ll.txt 2>&1
  echo "== live lists $ll $c"; grep -v amdgpu $O/stripe_batch_${c}_live_$ll.txt | grep -v "kernel class"
done; done
# stripes re-cut from BATCHED rank times (what bench.py --batch 4 does): does the slowest rank come down to the middle one's?
for c in c3 c4; do
  GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py $c --axis rows --batch 4 --in-flight 3,4 --ranks middle,slowest,edge --cuts-from-batch 4 > $O/stripe_batch_${c}_cuts_from_batch.txt 2>&1
  echo "== cuts from batched times $c"; grep -v amdgpu $O/stripe_batch_${c}_cuts_from_batch.txt | grep -v "kernel class"
done
unset GPU_MAX_HW_QUEUES
#   2b. review item 7: the compositor on a CU-masked stream of its own (diagnosis build: api.hip with -DGSPLAT_TEST_HOOKS),
#       two frames in flight — does leaving 32 / 64 / 96 compute units to the other frame's byte-bound kernels pay?
if [ -f build_variants/libgsplat_cumask.so ]; then
  for cus in none 224 192 160; do for c in c3 c4; do
    if [ $cus = none ]; then unset GSPLAT_PROBE_RENDER_CUS; else export GSPLAT_PROBE_RENDER_CUS=$cus; fi
    GSPLAT_LIB=$PWD/build_variants/libgsplat_cumask.so AB_SETTLE=64 timeout 300 python tools/ab_quick.py $c >> $O/ab_cumask.jsonl 2>> $O/ab_cumask.err
  done; done
  unset GSPLAT_PROBE_RENDER_CUS
  python - <<'PY'
import json
for l in open("gpurun_out/final_r06/ab_cumask.jsonl"):
    d = json.loads(l); print("cumask", d["config"], d["env"].get("GSPLAT_PROBE_RENDER_CUS", "-"), "one at a time", d["fps_one_at_a_time"], "two in flight", d["fps_two_in_flight"], "render ms", d["ms_kernel"]["render"])
PY
fi
if [ -z "${2:-}" ]; then tools/collect_final.sh r06 $GSPLAT_COMMIT > $O/collect.log 2>&1; echo "collect rc=$?" | tee -a $O/rc.txt; fi
cat $O/rc.txt
