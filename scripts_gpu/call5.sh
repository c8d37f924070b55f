#!/bin/bash
# round 4, GPU call 5: round B without the work it does not need — parity of the two-round tests, A/B against the build before
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_round or rounds or big_rect or overflow or relaid or stripe" > $O/pytest_rounds.log 2>&1; echo "rc=$?" >> $O/pytest_rounds.log
tail -4 $O/pytest_rounds.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "c3d or c3m or config5" > $O/pytest_full.log 2>&1; echo "rc=$?" >> $O/pytest_full.log
tail -4 $O/pytest_full.log
for cfg in c3d c3m c3; do
  for lib in build_variants/libgsplat_prev.so godotgaussiansplatting_amd/libgsplat_hip.so; do
    GSPLAT_LIB=$PWD/$lib timeout 240 python tools/ab_quick.py $cfg >> $O/ab.jsonl 2>> $O/ab.err
  done
done
cut -c1-560 $O/ab.jsonl
