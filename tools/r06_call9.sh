#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call9; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c3 --axis rows --batch 1,2,4 --in-flight 1,3 --ranks middle,slowest > $O/stripe_batch_c3.txt 2>&1; grep -v amdgpu $O/stripe_batch_c3.txt | grep "kernel class\|cuts"
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c4 --axis rows --batch 1,4 --in-flight 1,3 --ranks slowest > $O/stripe_batch_c4.txt 2>&1; grep -v amdgpu $O/stripe_batch_c4.txt | grep "kernel class\|cuts"
unset GPU_MAX_HW_QUEUES
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bench_ or stand_in_rccl or batched" > $O/parity.txt 2>&1; echo "parity rc=$?"; tail -15 $O/parity.txt
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 --dist group --finalize on --no-cpu-baseline > $O/bench_c3_force_dist_batch4.json 2> $O/err1.txt; tail -3 $O/err1.txt
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 --dist group --finalize on --no-cpu-baseline --batch 1 > $O/bench_c3_force_dist_batch1.json 2> $O/err2.txt; tail -3 $O/err2.txt
python - <<'PY'
import json
for b in (4,1):
    d=json.load(open(f"gpurun_out/call9/bench_c3_force_dist_batch{b}.json"))
    print("forced one-rank group path, batch", b, "value", round(d["value"],1), "frame_equal", d["frame_equal"], "stage", d["dist_stage"], d["config"]["frames_in_flight"])
PY
