#!/bin/bash
# round 4, GPU call 3: suite after the memory diet, memory figures, read-back engine probe, stripe models
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for cfg in c3 c4; do
  for lib in build_variants/libgsplat_base.so godotgaussiansplatting_amd/libgsplat_hip.so; do
    GSPLAT_LIB=$PWD/$lib timeout 240 python tools/ab_quick.py $cfg >> $O/ab.jsonl 2>> $O/ab.err
  done
done
cut -c1-330 $O/ab.jsonl
for cfg in c3 c5; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline --steps 60 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
python - <<'PY'
import json
for f in ("bench_c3","bench_c5"):
    try:
        d=json.loads(open("gpurun_out/c3/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d.get("sequential_fps"), d["scene_stats"]["device_bytes"]/1e9, d["scene_stats"]["device_bytes_scene"]/1e9, d["scene_stats"]["device_bytes_per_frame_in_flight"]/1e9, d["pairs_round"])
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 200 python tools/d2h_probe.py c3 >> $O/d2h.jsonl 2>> $O/d2h.err
HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0 timeout 200 python tools/d2h_probe.py c3 >> $O/d2h.jsonl 2>> $O/d2h.err
GPU_BLIT_ENGINE_TYPE=1 timeout 200 python tools/d2h_probe.py c3 >> $O/d2h.jsonl 2>> $O/d2h.err
GPU_BLIT_ENGINE_TYPE=2 timeout 200 python tools/d2h_probe.py c3 >> $O/d2h.jsonl 2>> $O/d2h.err
HSA_ENABLE_SDMA=0 timeout 200 python tools/d2h_probe.py c3 >> $O/d2h.jsonl 2>> $O/d2h.err
cat $O/d2h.jsonl
for c in c3 c4; do GSPLAT_ROUNDS=off timeout 400 python tools/stripe_model.py $c cull > $O/stripe_model_$c.txt 2>&1; tail -12 $O/stripe_model_$c.txt; done
du -sh gpurun_out
