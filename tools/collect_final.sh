#!/bin/bash
# Everything profiles/ holds for the final build of a round, in ONE gpurun call, all from the same tree:
#   tools/collect_final.sh <round tag, e.g. r04> <commit the tree was taken from>
# Order matters: the PMC passes come first (they rewrite profiles/pmc_traffic.json and profiles/sq_bound.json, stamped with
# the commit), the bench lines after them read the fresh files for roofline.traffic / roofline.binding_bound.  Runs on the
# GPU box; only summaries leave it (gpurun merges at most 64 MiB back): the rocpd databases are summarised in place and
# removed.  Copy gpurun_out/final/* to profiles/.
R=${1:-r05}; export GSPLAT_COMMIT=${2:-unknown}
cd $GRAFT_REPO_ROOT
F=gpurun_out/final; mkdir -p $F
FAIL=0
echo "$R $GSPLAT_COMMIT $(date -u +%FT%TZ)" > $F/${R}_stamp.txt
prof() {  # <config> <GSPLAT_ROUNDS setting the context settles on in a plain run>
  tools/profile_gpu.sh $1 final/prof_$1 $2 > /dev/null 2>&1
  if ! python tools/summarize_profile.py $F/prof_$1 $F/${R}_$1 $1 > $F/summarize_$1.log 2>&1; then
    # a --pmc pass that did not collect (a counter zero for every kernel) is refused by the summariser: once more, the counter
    # passes alone; a second refusal is recorded and fails the collection (no entry in pmc_traffic.json, "traffic": null)
    cp $F/prof_$1/pmc_fetch.err $F/${R}_$1_pmc_fetch_first_try.err 2>/dev/null; cp $F/prof_$1/pmc_write.err $F/${R}_$1_pmc_write_first_try.err 2>/dev/null
    PROFILE_ONLY_PMC=1 tools/profile_gpu.sh $1 final/prof_$1 $2 > /dev/null 2>&1
    if ! python tools/summarize_profile.py $F/prof_$1 $F/${R}_$1 $1 > $F/summarize_$1.log 2>&1; then
      echo "$1: $(tail -1 $F/summarize_$1.log)" >> $F/${R}_collection_failures.txt; FAIL=1
    fi
  fi
  rm -rf $F/prof_$1
}
cp profiles/pmc_traffic.json $F/pmc_traffic.json 2>/dev/null   # summarize_profile.py merges into the copy next to its prefix
prof c3 off; prof c3m 0.25; prof c3d 0.011; prof c3r 0.032; prof c4 off; prof c5 off
cp $F/pmc_traffic.json profiles/pmc_traffic.json
# which limit binds the compositor (SQ counters -> profiles/sq_bound.json, read by bench.py)
for c in c3 c4; do timeout 400 python tools/sq_bound.py $c $GSPLAT_COMMIT > $F/sq_bound_$c.txt 2>&1; done
rm -rf gpurun_out/pmc_one
cp profiles/sq_bound.json $F/sq_bound.json
for c in c3 c3m c3d c3r c1 c2 c4 c5; do timeout 400 python bench.py --config $c > $F/${R}_bench_$c.json 2> $F/bench_$c.err; done
timeout 300 python bench.py --config c3 --camera orbit --no-cpu-baseline > $F/${R}_bench_c3_orbit.json 2> $F/bench_c3_orbit.err
timeout 300 python bench.py --config c2 --while-loading --no-cpu-baseline > $F/${R}_bench_c2_while_loading.json 2> $F/bench_c2_loading.err
# the product's multi-GPU path with one rank (all this box has): gsplat_group_*, and the torch.distributed host for A/B
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 --dist group --no-cpu-baseline > $F/${R}_bench_c3_force_dist_group.json 2> $F/force_dist_group.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 --dist torch --no-cpu-baseline > $F/${R}_bench_c3_force_dist_torch.json 2> $F/force_dist_torch.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 --dist group --finalize on --no-cpu-baseline > $F/${R}_bench_c3_force_dist_group_morton_cull.json 2> $F/force_dist_group_cull.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 --dist group --finalize on --no-cpu-baseline --batch 1 > $F/${R}_bench_c3_force_dist_group_morton_cull_batch1.json 2> $F/force_dist_group_cull_b1.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c2 --dist group --finalize on --no-cpu-baseline > $F/${R}_bench_c2_force_dist_group_oracle_check.json 2> $F/force_dist_group_c2.err
# the opt-in hardware exp2 (GSPLAT_FLAG_FAST_EXP: not the contract, never `value`)
timeout 300 python bench.py --config c3 --fast-exp --no-cpu-baseline > $F/${R}_bench_c3_fast_exp.json 2> $F/bench_c3_fast_exp.err
# a scene from a file (bench.py --ply): c2's rows written as an INRIA .ply, read back by PlyFile.parse
python -c "
import sys; sys.path.insert(0, '.')
from godotgaussiansplatting_amd import scenes
scenes.write_ply('/tmp/c2_rows.ply', scenes.config_rows('c2'))"
timeout 300 python bench.py --ply /tmp/c2_rows.ply > $F/${R}_bench_ply_c2_rows.json 2> $F/bench_ply.err
# stripe ranks as bench.py --gpus N runs them: Morton layout, block culling, equal keys in storage order (GSPLAT_FLAG_TIES_STORAGE_ORDER)
# (round 6: ROW stripes as the bench cuts them, B frames per launch sequence x R sequences in flight, eight hardware queues)
for c in c3 c4; do GPU_MAX_HW_QUEUES=8 GSPLAT_ROUNDS=off timeout 600 python tools/stripe_batch.py $c --axis rows --batch 1,2,4 > $F/${R}_stripe_batch_${c}_rows_8queues.txt 2>&1; done
for c in c3 c4; do GSPLAT_ROUNDS=off timeout 200 python tools/stripe_kernels.py $c 8 3 ties > $F/${R}_stripe_kernels_$c.txt 2>&1; done
# which launches of the projection kernel are slow, and what shares the chip with them (per-call trace of the default command)
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
for c in c3 c4; do
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $REPO/$F/calls_$c -o calls -- python $REPO/bench.py --config $c --no-cpu-baseline > /dev/null 2> $REPO/$F/calls_$c.err
  python $REPO/tools/outliers.py $REPO/$F/calls_$c project_kernel 1.5 > $REPO/$F/${R}_project_outliers_$c.txt 2>&1
  rm -rf $REPO/$F/calls_$c
done
cd $REPO
timeout 200 python tools/d2h_probe.py c3 > $F/${R}_d2h_probe.jsonl 2>> $F/d2h.err
timeout 200 python tools/d2h_probe.py c4 >> $F/${R}_d2h_probe.jsonl 2>> $F/d2h.err
# where a compositor wave spends its time (needs the probe build: build_variants/libgsplat_tl.so = raster.hip with -DGS_PROBE_TIMELINE)
if [ -f build_variants/libgsplat_tl.so ]; then
  for c in c3 c4; do GSPLAT_LIB=$PWD/build_variants/libgsplat_tl.so GSPLAT_ROUNDS=off timeout 300 python tools/render_timeline.py $c > $F/${R}_render_timeline_$c.txt 2>&1; done
fi
cp gpurun_out/twin_report_*.json $F/ 2>/dev/null
du -sh gpurun_out; ls $F
if [ "$FAIL" != 0 ]; then echo "COLLECTION FAILED:"; cat $F/${R}_collection_failures.txt; fi
exit $FAIL
