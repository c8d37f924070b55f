#!/bin/bash
# round 6, call 1: the new parity tests (multi-GPU default configuration at workload size, the stand-in RCCL session with the
# bench's real flags, the staged bench fallback) and the counter passes of c3r / c3d that round 5 summarised empty
cd $GRAFT_REPO_ROOT
O=gpurun_out/call1; mkdir -p $O
export GSPLAT_COMMIT=${1:-unknown}
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -k "multi_gpu_default" -s > $O/full_size_multi.txt 2>&1; echo "full_size rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stand_in_rccl or bench_ or atomic_and_ballot or opt_in or one_pass" > $O/parity_subset.txt 2>&1; echo "parity rc=$?" >> $O/rc.txt
F=gpurun_out/final; mkdir -p $F; R=r06
cp profiles/pmc_traffic.json $F/pmc_traffic.json
for cr in "c3r 0.032" "c3d 0.011"; do
  set -- $cr
  tools/profile_gpu.sh $1 final/prof_$1 $2 > /dev/null 2>&1
  python tools/summarize_profile.py $F/prof_$1 $F/${R}_$1 $1 > $F/summarize_$1.log 2>&1; echo "summarize $1 rc=$?" >> $O/rc.txt
  cp $F/prof_$1/pmc_write.err $O/pmc_write_$1.err; cp $F/prof_$1/pmc_fetch.err $O/pmc_fetch_$1.err
  rm -rf $F/prof_$1
done
cat $O/rc.txt; tail -5 $O/full_size_multi.txt; tail -5 $O/parity_subset.txt
