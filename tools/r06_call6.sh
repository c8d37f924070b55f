#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" GSPLAT_ROUNDS=off timeout 300 python tools/inflight_probe.py $V 2>&1 | grep -v amdgpu; }
V=notorch_nonull run X=1
V=notorch run GPU_MAX_HW_QUEUES=8
V=notorch run GPU_MAX_HW_QUEUES=2
V=notorch_dummyuse run PROBE_DUMMY_STREAMS=1
V=notorch_dummyuse run PROBE_DUMMY_STREAMS=2
V=notorch_dummyuse run PROBE_DUMMY_STREAMS=3
V=notorch run PROBE_DUMMY_STREAMS=2
