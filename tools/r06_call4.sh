#!/bin/bash
# round 6, call 4: rows vs columns for the stripe ranks, with batches (the bench cuts ROWS, round 5's model measured COLUMNS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/call4; mkdir -p $O
for ax in columns rows; do
  GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c3 --axis $ax --ranks middle,slowest --fixed > $O/stripe_batch_c3_${ax}_fixed.txt 2>&1
  grep -v amdgpu $O/stripe_batch_c3_${ax}_fixed.txt
done
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c3 --axis columns --ranks middle,slowest > $O/stripe_batch_c3_columns.txt 2>&1; grep -v amdgpu $O/stripe_batch_c3_columns.txt
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c4 --axis columns --batch 1,2 --ranks middle,slowest > $O/stripe_batch_c4_columns.txt 2>&1; grep -v amdgpu $O/stripe_batch_c4_columns.txt
GSPLAT_ROUNDS=off STRIPE_MODEL_G=8 timeout 500 python tools/stripe_model.py c3 cull+ties > $O/stripe_model_c3_old_tool.txt 2>&1; grep -v amdgpu $O/stripe_model_c3_old_tool.txt | tail -12
