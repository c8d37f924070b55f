#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "block_cull or batched or finalized or storage_order or stand_in_rccl or degenerate or uploads" > $O/parity.txt 2>&1; echo "parity rc=$?"; tail -5 $O/parity.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -k "multi_gpu_default" > $O/full.txt 2>&1; echo "full rc=$?"; tail -3 $O/full.txt
export GPU_MAX_HW_QUEUES=8
for c in c3 c4; do
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py $c --axis rows --batch 1,2,4 --in-flight 1,3,4 --ranks middle,slowest > $O/stripe_batch_$c.txt 2>&1; grep -v amdgpu $O/stripe_batch_$c.txt
done
