#!/bin/bash
# Everything profiles/ holds for the final build of a round, in one gpurun call.  Runs on the GPU box; only summaries
# leave it (gpurun merges at most 64 MiB back): the rocpd databases are summarised in place and removed.
cd $GRAFT_REPO_ROOT
F=gpurun_out/final; mkdir -p $F
for c in c3 c3d c1 c2 c4 c5; do python bench.py --config $c > $F/r02_bench_$c.json 2> $F/bench_$c.err; done
GSPLAT_FORCE_DIST=1 python bench.py --config c3 > $F/r02_bench_c3_force_dist.json 2> $F/force_dist.err
cp profiles/pmc_traffic.json $F/pmc_traffic_before.json
prof() {  # <config> <GSPLAT_ROUNDS setting the context settles on in a plain run>
  tools/profile_gpu.sh $1 final/prof_$1 $2 > /dev/null 2>&1
  python tools/summarize_profile.py $F/prof_$1 $F/r02_$1 $1 > $F/summarize_$1.log 2>&1
  rm -rf $F/prof_$1
}
prof c3 off; prof c3d 0.011; prof c4 off; prof c5 0.2
cp profiles/pmc_traffic.json $F/pmc_traffic.json
for c in c3 c4 c5; do GSPLAT_ROUNDS=off python tools/stripe_model.py $c cull > $F/r02_stripe_model_$c.txt 2>&1; done
tools/pmc_one.sh c3 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" render > $F/sq1.txt 2>&1
tools/pmc_one.sh c3 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" render > $F/sq2.txt 2>&1
rm -rf gpurun_out/pmc_one
du -sh gpurun_out
