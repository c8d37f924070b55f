#!/bin/bash
# The part of tools/collect_final.sh that the driver's line and the roofline are read from, for a late change of ONE kernel
# when the round's GPU minutes no longer cover the whole collection:
#   tools/collect_core.sh <round tag> <commit>
# kernel stats + PMC traffic of c3 and c4, the compositor's SQ counters, every configuration's bench line and the orbit
# line; then a parity subset.  Writes gpurun_out/final/<tag>_stamp_core.txt; the files it does not produce keep the
# stamp of the full collection (profiles/README.md says which is which).
R=${1:-r04}; export GSPLAT_COMMIT=${2:-unknown}
cd $GRAFT_REPO_ROOT
F=gpurun_out/final; mkdir -p $F
FAIL=0
echo "$R $GSPLAT_COMMIT $(date -u +%FT%TZ) core" > $F/${R}_stamp_core.txt
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
cp profiles/pmc_traffic.json $F/pmc_traffic.json 2>/dev/null
prof c3 off; prof c4 off
cp $F/pmc_traffic.json profiles/pmc_traffic.json
for c in c3 c4; do timeout 300 python tools/sq_bound.py $c $GSPLAT_COMMIT > $F/sq_bound_$c.txt 2>&1; done
rm -rf gpurun_out/pmc_one
cp profiles/sq_bound.json $F/sq_bound.json
for c in c3 c4 c3m c3d c1 c2 c5; do timeout 300 python bench.py --config $c > $F/${R}_bench_$c.json 2> $F/bench_$c.err; done
timeout 200 python bench.py --config c3 --camera orbit --no-cpu-baseline > $F/${R}_bench_c3_orbit.json 2> $F/bench_c3_orbit.err
date -u +%T > $F/core_done.txt
timeout 170 python -m pytest tests/test_gpu_parity.py -x -q -k "image or round or two" 2>&1 | tail -3 > $F/core_parity_subset.txt
ls $F
if [ "$FAIL" != 0 ]; then echo "COLLECTION FAILED:"; cat $F/${R}_collection_failures.txt; fi
exit $FAIL
