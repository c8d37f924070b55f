#!/bin/bash
# round 4, GPU call 2: full GPU suite (whole-frame twin checks, c3m), A/B vs round 3, bench lines (default, group, torch, ply), outliers trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -16 $O/pytest.log
for cfg in c3 c1 c3d c4 c2; do
  for lib in build_variants/libgsplat_base.so godotgaussiansplatting_amd/libgsplat_hip.so; do
    GSPLAT_LIB=$PWD/$lib timeout 240 python tools/ab_quick.py $cfg >> $O/ab.jsonl 2>> $O/ab.err
  done
done
timeout 240 python tools/ab_quick.py c3m >> $O/ab.jsonl 2>> $O/ab.err
GSPLAT_ROUNDS=off timeout 240 python tools/ab_quick.py c3m >> $O/ab.jsonl 2>> $O/ab.err
cat $O/ab.jsonl | cut -c1-420
timeout 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_c3_group1.json 2> $O/bench_c3_group1.err; tail -3 $O/bench_c3_group1.err
GSPLAT_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --dist torch > $O/bench_c3_torch1.json 2> $O/bench_c3_torch1.err; tail -3 $O/bench_c3_torch1.err
python -c "
import sys; sys.path.insert(0,'.')
from godotgaussiansplatting_amd import scenes
scenes.write_ply('/tmp/c2_like.ply', scenes.synthetic_rows(1_000_000, 2, 0))
"
timeout 300 python bench.py --ply /tmp/c2_like.ply > $O/bench_ply.json 2> $O/bench_ply.err; tail -3 $O/bench_ply.err
timeout 300 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
python - <<'PY'
import json
for f in ("bench_c3","bench_c3_group1","bench_c3_torch1","bench_ply","bench_c2"):
    try:
        d=json.loads(open("gpurun_out/c2/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d.get("sequential_fps"), d.get("value_moving_camera"), d["config"]["workload"][:60], d.get("dist"), d.get("ms_gather"), (d.get("roofline") or {}).get("frac"), ((d.get("roofline") or {}).get("binding_bound") or {}).get("frac"), (d.get("parity_check") or {}).get("rgba_bit_exact"))
    except Exception as e:
        print(f, "FAILED", e)
PY
# per-call trace: which bench leg owns the slow project_kernel launches?
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
for cfg in c3 c4; do
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $REPO/$O/calls_$cfg -o calls -- python $REPO/bench.py --config $cfg --no-cpu-baseline > /dev/null 2> $REPO/$O/calls_$cfg.err
  python $REPO/tools/outliers.py $REPO/$O/calls_$cfg project_kernel 1.5 > $REPO/$O/outliers_$cfg.txt 2>&1
  rm -rf $REPO/$O/calls_$cfg
done
cd $REPO
cat $O/outliers_c3.txt | head -40
du -sh gpurun_out
