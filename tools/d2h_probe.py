#!/usr/bin/env python3
"""How the pipelined read-back (gsplat_render_async + gsplat_readback_wait) shares the chip with the next frame:
frames/s with every frame copied to pinned host memory, and the GPU time of a frame's first kernel while a copy runs.
Run under different runtime settings (the copy engine is the HIP runtime's choice, read at process start), e.g.
    python tools/d2h_probe.py c3
    HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0 python tools/d2h_probe.py c3
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from godotgaussiansplatting_amd import capi, scenes  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
ctx = capi.Context(n, w, h)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
fr = capi.make_frame(vp, cam)
for _ in range(30):
    ctx.render(fr)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ctx.render(fr)
ctx.synchronize()
plain = 200 / (time.perf_counter() - t0)
prev = None
for _ in range(8):
    tk = ctx.render_async(fr)
    if prev is not None:
        ctx.readback_wait(prev)
    prev = tk
t0 = time.perf_counter()
for _ in range(200):
    tk = ctx.render_async(fr)
    ctx.readback_wait(prev)
    prev = tk
ctx.readback_wait(prev)
d2h = 200 / (time.perf_counter() - t0)
ctx.set_timing(capi.FLAG_TIMING)
tk = ctx.render_async(fr)
ctx.readback_wait(tk)
st = ctx.stats()
print(json.dumps({"config": cfg, "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_FORCE_BLIT_COPY_SIZE", "GPU_BLIT_ENGINE_TYPE", "GSPLAT_READBACK", "GSPLAT_READBACK_WGS") if os.environ.get(k)},
                  "fps_one_at_a_time": round(plain, 1), "fps_with_d2h": round(d2h, 1), "ms_readback": round(st["ms_readback"], 3),
                  "frame_ms_gpu": round(st["ms_total"], 3), "image_MB": w * h * 16 / 1e6}))
ctx.close()
