#!/usr/bin/env python3
"""Frames delivered to pinned host memory with TWO contexts in flight, each with a read-back ring of its own
(gsplat_render_async on views of one scene, alternating): does the pipelined host rate follow `value` (two frames in
flight) instead of the one-at-a-time rate?   usage: python tools/d2h_two_contexts.py <config> [rgb]"""
import os
import sys
import time
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
for R in (1, 2, 3):
    ring = [ctx] + [ctx.view(flags=RGB) for _ in range(R - 1)]
    for c in ring:
        for _ in range(64):
            c.render(fr); c.synchronize()
    pend = [None] * R
    def step(k):
        c = ring[k % R]
        tk = c.render_async(fr)
        if pend[k % R] is not None:
            c.readback_wait(pend[k % R])
        pend[k % R] = tk
    for k in range(6 * R):
        step(k)
    nfr = 200
    t0 = time.perf_counter()
    for k in range(nfr):
        step(k)
    for i, c in enumerate(ring):
        c.readback_wait(pend[i]); pend[i] = None
    dt = time.perf_counter() - t0
    print(f"{cfg} {'rgb' if RGB else 'rgba'} contexts in flight {R}: {nfr / dt:.0f} frames/s delivered to host memory")
    for c in ring[1:]:
        c.close()
ctx.close()
