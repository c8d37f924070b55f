#!/bin/bash
# round 6, call 2: geometry-eager lazy frames (GSPLAT_GEO=on) — parity first, then the A/B against the recomputing compositor
# back to back (c3 / c3r / c4, A B A B), the compositor's WRITE_SIZE / FETCH_SIZE with it, the workload-size multi-GPU
# configuration tests, and c3r's counter passes once more
cd $GRAFT_REPO_ROOT
O=gpurun_out/call2; mkdir -p $O
export GSPLAT_COMMIT=${1:-unknown}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "geometry_eager or opt_in or pick or two_round_frames_are" > $O/parity_geo.txt 2>&1; echo "parity_geo rc=$?" >> $O/rc.txt
for rep in 1 2; do for c in c3 c3r c4; do for g in off on; do
  GSPLAT_GEO=$g timeout 300 python tools/ab_quick.py $c >> $O/ab_geo.jsonl 2>> $O/ab_geo.err
done; done; done
for g in off on; do
  GSPLAT_GEO=$g tools/pmc_one.sh c3 "WRITE_SIZE" render_kernel > $O/pmc_write_c3_geo_$g.txt 2>&1
  GSPLAT_GEO=$g tools/pmc_one.sh c3 "FETCH_SIZE" render_kernel > $O/pmc_fetch_c3_geo_$g.txt 2>&1
  GSPLAT_GEO=$g tools/pmc_one.sh c3 "WRITE_SIZE" project_kernel > $O/pmc_write_c3_proj_geo_$g.txt 2>&1
done
rm -rf gpurun_out/pmc_one
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -k "multi_gpu_default" -s > $O/full_size_multi.txt 2>&1; echo "full_size rc=$?" >> $O/rc.txt
F=gpurun_out/final; mkdir -p $F; R=r06
cp profiles/pmc_traffic.json $F/pmc_traffic.json
tools/profile_gpu.sh c3r final/prof_c3r 0.032 > /dev/null 2>&1
python tools/summarize_profile.py $F/prof_c3r $F/${R}_c3r c3r > $F/summarize_c3r.log 2>&1; echo "summarize c3r rc=$?" >> $O/rc.txt
cp $F/prof_c3r/pmc_tries.txt $O/pmc_tries_c3r.txt
rm -rf $F/prof_c3r
cat $O/rc.txt; tail -3 $O/parity_geo.txt; tail -3 $O/full_size_multi.txt; cat $O/ab_geo.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['env'].get('GSPLAT_GEO'), d['fps_one_at_a_time'], d['fps_two_in_flight'], d['ms_kernel']['render'], d['ms_kernel']['project'], d['ms_pass']['total'])
"
