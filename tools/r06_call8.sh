#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call8; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
GSPLAT_ROUNDS=off timeout 300 python tools/inflight_probe.py torch_first 2>&1 | grep -v amdgpu
for c in c3 c4; do
  GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py $c --axis rows --batch 1,2,3,4 --ranks middle,slowest > $O/stripe_batch_${c}_rows_q8.txt 2>&1
  grep -v amdgpu $O/stripe_batch_${c}_rows_q8.txt; cp gpurun_out/stripe_batch_$c.json $O/stripe_batch_${c}_rows_q8.json
done
export GPU_MAX_HW_QUEUES=16
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c3 --axis rows --batch 1,4 --in-flight 2,4,6 --ranks middle > $O/stripe_batch_c3_rows_q16.txt 2>&1; grep -v amdgpu $O/stripe_batch_c3_rows_q16.txt
