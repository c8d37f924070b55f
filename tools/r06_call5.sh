#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/call5; mkdir -p $O
for c in c3 c4; do
  GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py $c --axis rows --batch 1,2,4 --ranks middle,slowest > $O/stripe_batch_${c}_rows.txt 2>&1
  grep -v amdgpu $O/stripe_batch_${c}_rows.txt
done
GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c3 --axis columns --batch 1,2,4 --ranks middle,slowest > $O/stripe_batch_c3_columns.txt 2>&1; grep -v amdgpu $O/stripe_batch_c3_columns.txt
for ps in split wide; do GSPLAT_PAIR_SORT=$ps GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py c3 --axis rows --batch 4 --in-flight 1,3 --ranks middle > $O/stripe_batch_c3_rows_pairsort_$ps.txt 2>&1; grep -v amdgpu $O/stripe_batch_c3_rows_pairsort_$ps.txt; done
