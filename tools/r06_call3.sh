#!/bin/bash
# round 6, call 3: batched frames — parity first (every batched frame array_equal the unbatched one and the oracle's), then the
# stripe-rank model with batches of 1-4 frames and 1-3 batches in flight (c3, c4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/call3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batched" > $O/parity_batch.txt 2>&1; echo "parity_batch rc=$?" >> $O/rc.txt
tail -30 $O/parity_batch.txt
for c in c3 c4; do GSPLAT_ROUNDS=off timeout 900 python tools/stripe_batch.py $c > $O/stripe_batch_$c.txt 2>&1; cp gpurun_out/stripe_batch_$c.json $O/ 2>/dev/null; tail -40 $O/stripe_batch_$c.txt; done
cat $O/rc.txt
