#!/bin/bash
# round 4, GPU call 4: suite after the slot-on-demand change, memory of band-0 scenes, default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for cfg in c2 c5; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline --steps 60 --frames-in-flight 1 > $O/bench_${cfg}_1.json 2> $O/bench_${cfg}_1.err
done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_c3_k20.json 2> $O/bench_c3_k20.err
python - <<'PY'
import json
for f in ("bench_c2_1","bench_c5_1","bench_c3_k20"):
    try:
        d=json.loads(open("gpurun_out/c4/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d.get("sequential_fps"), d.get("value_moving_camera"), (d.get("orbit") or {}).get("vs_fixed_camera"), d["scene_stats"]["device_bytes"]/1e9, d["scene_stats"]["device_bytes_scene"]/1e9, d["scene_stats"]["device_bytes_per_frame_in_flight"]/1e9, (d.get("parity_check") or {}).get("rgba_bit_exact"))
    except Exception as e:
        print(f, "FAILED", e)
PY
for cfg in c2 c1; do
  for lib in build_variants/libgsplat_base.so godotgaussiansplatting_amd/libgsplat_hip.so; do
    GSPLAT_LIB=$PWD/$lib timeout 240 python tools/ab_quick.py $cfg >> $O/ab.jsonl 2>> $O/ab.err
  done
done
cut -c1-200 $O/ab.jsonl
