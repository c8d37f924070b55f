#!/bin/bash
# round 4, GPU call 1: parity of the fused build, stress of the stripe + block-cull test, A/B against round 3's library
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
if grep -q " passed" $O/pytest.log && ! grep -q "failed" $O/pytest.log; then
  # the test that did not finish under round 3's last experiment (projection blocks XCD by XCD): 6 x all four variants
  for i in 1 2 3 4 5 6; do
    timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_block_cull_stripes_with_last_tile_exchange" >> $O/stress.log 2>&1
    echo "iter $i rc=$?" >> $O/stress.log
  done
  grep "rc=\|passed\|failed" $O/stress.log | tail -14
fi
for cfg in c3 c1 c3d c4; do
  for lib in build_variants/libgsplat_base.so godotgaussiansplatting_amd/libgsplat_hip.so; do
    GSPLAT_LIB=$PWD/$lib timeout 240 python tools/ab_quick.py $cfg >> $O/ab.jsonl 2>> $O/ab.err
  done
done
cat $O/ab.jsonl
# kernel trace of the new build (one frame at a time)
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/$O/trace -o trace -- python $REPO/bench.py --config c3 --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --frames-in-flight 1 > $REPO/$O/bench_trace.json 2> $REPO/$O/trace.err
cd $REPO
python tools/summarize_profile.py $O $O/c3 c3 > $O/summarize.log 2>&1
find $O/trace -name "*.db" -delete 2>/dev/null; find $O/trace -size +5M -delete 2>/dev/null
cat $O/c3_kernel_stats.md 2>/dev/null | head -40
du -sh gpurun_out
