#!/bin/bash
# Everything profiles/ holds for the final build of a round, in ONE gpurun call, all from the same tree:
#   tools/collect_final.sh <round tag, e.g. r03> <commit the tree was taken from>
# Order matters: the PMC passes come first (they rewrite profiles/pmc_traffic.json, stamped with the commit), the bench
# lines after them read the fresh file for roofline.traffic.  Runs on the GPU box; only summaries leave it (gpurun merges
# at most 64 MiB back): the rocpd databases are summarised in place and removed.  Copy gpurun_out/final/* to profiles/.
R=${1:-r03}; export GSPLAT_COMMIT=${2:-unknown}
cd $GRAFT_REPO_ROOT
F=gpurun_out/final; mkdir -p $F
echo "$R $GSPLAT_COMMIT $(date -u +%FT%TZ)" > $F/${R}_stamp.txt
prof() {  # <config> <GSPLAT_ROUNDS setting the context settles on in a plain run>
  tools/profile_gpu.sh $1 final/prof_$1 $2 > /dev/null 2>&1
  python tools/summarize_profile.py $F/prof_$1 $F/${R}_$1 $1 > $F/summarize_$1.log 2>&1
  rm -rf $F/prof_$1
}
cp profiles/pmc_traffic.json $F/pmc_traffic.json 2>/dev/null   # summarize_profile.py merges into the copy next to its prefix
prof c3 off; prof c3d 0.011; prof c4 off; prof c5 0.2
cp $F/pmc_traffic.json profiles/pmc_traffic.json
for c in c3 c3d c1 c2 c4 c5; do timeout 400 python bench.py --config $c > $F/${R}_bench_$c.json 2> $F/bench_$c.err; done
timeout 300 python bench.py --config c3 --camera orbit --no-cpu-baseline > $F/${R}_bench_c3_orbit.json 2> $F/bench_c3_orbit.err
timeout 300 python bench.py --config c2 --while-loading --no-cpu-baseline > $F/${R}_bench_c2_while_loading.json 2> $F/bench_c2_loading.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --config c3 > $F/${R}_bench_c3_force_dist.json 2> $F/force_dist.err
for c in c3 c4; do GSPLAT_ROUNDS=off timeout 400 python tools/stripe_model.py $c cull > $F/${R}_stripe_model_$c.txt 2>&1; done
tools/pmc_one.sh c3 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" render > $F/sq1.txt 2>&1
tools/pmc_one.sh c3 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" render > $F/sq2.txt 2>&1
rm -rf gpurun_out/pmc_one
# where a compositor wave spends its time (needs the probe build: _ab/variant.sh tl raster -DGS_PROBE_TIMELINE)
if [ -f build_variants/libgsplat_tl.so ]; then
  for c in c3 c4; do GSPLAT_LIB=$PWD/build_variants/libgsplat_tl.so GSPLAT_ROUNDS=off timeout 300 python tools/render_timeline.py $c > $F/${R}_render_timeline_$c.txt 2>&1; done
fi
# instruction issue costs behind the blend loop's form (tools/valu_rates.hip, tools/step_rates.hip; built by hipcc -O2)
[ -x tools/valu_rates ] && timeout 250 tools/valu_rates > $F/valu_rates.txt 2>&1
[ -x tools/step_rates ] && timeout 250 tools/step_rates > $F/step_rates.txt 2>&1
cp gpurun_out/twin_report_*.json $F/ 2>/dev/null
du -sh gpurun_out; ls $F
