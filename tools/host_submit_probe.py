import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes
cfg = sys.argv[1]
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
FLAGS = capi.FLAG_BLOCK_CULL | capi.FLAG_TIES_STORAGE_ORDER
ctx = capi.Context(n, w, h, flags=FLAGS)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
ctx.finalize_scene()
fr = capi.make_frame(vp, cam)
import torch
TOP = torch.zeros(1, dtype=torch.int32, device="cuda")
ctx.render_begin(fr, TOP.data_ptr()); ctx.render_end(); ctx.synchronize()
cuts = {"c3": (55, 60), "c4": (109, 120)}[cfg]
for R in (1, 2, 3, 4, 6):
    ring = [ctx.view(stripe=(capi.STRIPE_COLUMNS, cuts[0], cuts[1]), flags=FLAGS) for _ in range(R)]
    def render(c):
        c.render_begin(fr); c.render_end(frame_last_tile_ptr=TOP.data_ptr())
    for k in range(3 * R): render(ring[k % R])
    for c in ring: c.synchronize()
    reps = 120
    host = 0.0
    t0 = time.perf_counter()
    for k in range(reps):
        a = time.perf_counter(); render(ring[k % R]); host += time.perf_counter() - a
    t_sub = time.perf_counter() - t0
    for c in ring: c.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"{cfg} R={R}: {dt:.3f} ms/frame; host time inside render calls {host/reps*1e3:.3f} ms/frame; loop returned after {t_sub/reps*1e3:.3f} ms/frame")
    for c in ring: c.close()
