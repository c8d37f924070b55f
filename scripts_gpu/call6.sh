#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_round or rounds or overflow or relaid" > $O/pytest_rounds.log 2>&1; echo "rc=$?" >> $O/pytest_rounds.log
tail -3 $O/pytest_rounds.log
for rep in 1 2; do
for cfg in c3d c3m; do
  for lib in build_variants/libgsplat_prev.so godotgaussiansplatting_amd/libgsplat_hip.so; do
    GSPLAT_LIB=$PWD/$lib AB_SETTLE=100 timeout 240 python tools/ab_quick.py $cfg >> $O/ab.jsonl 2>> $O/ab.err
  done
done
done
python - <<'PY'
import json
for l in open("gpurun_out/c6/ab.jsonl"):
    d=json.loads(l); k=d["ms_kernel"]
    print(d["lib"][10:14], d["config"], d["fps_one_at_a_time"], d["fps_two_in_flight"], d["ms_pass"]["total"], "scan",k["scan"],"emit",k["emit"],"ds",k["sort_downsweep"],"render",k["render"], d["pairs_round"])
PY
