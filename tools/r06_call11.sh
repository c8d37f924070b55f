#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/live_blocks_probe.py c3 2>&1 | grep -v amdgpu
python tools/live_blocks_probe.py c4 2>&1 | grep -v amdgpu
