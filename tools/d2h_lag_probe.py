#!/usr/bin/env python3
"""The pinned read-back ring with the host waiting for frame k - LAG instead of k - 1 after submitting frame k (the ring
holds three host images: LAG may be 1 or 2).   usage: python tools/d2h_lag_probe.py <config> [rgb]"""
import os
import sys
import time
from collections import deque
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from godotgaussiansplatting_amd import capi, scenes

cfg = sys.argv[1]
RGB = capi.FLAG_READBACK_RGB if len(sys.argv) > 2 and sys.argv[2] == "rgb" else 0
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
ctx = capi.Context(n, w, h, flags=RGB)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
fr = capi.make_frame(vp, cam)
for _ in range(64):
    ctx.render(fr); ctx.synchronize()
for lag in (1, 2, 1, 2):
    pend = deque()
    def step():
        pend.append(ctx.render_async(fr))
        if len(pend) > lag:
            ctx.readback_wait(pend.popleft())
    for _ in range(12):
        step()
    nfr = 300
    t0 = time.perf_counter()
    for _ in range(nfr):
        step()
    while pend:
        ctx.readback_wait(pend.popleft())
    dt = time.perf_counter() - t0
    print(f"{cfg} {'rgb' if RGB else 'rgba'} host waits for frame k-{lag}: {nfr / dt:.0f} frames/s delivered")
ctx.close()
