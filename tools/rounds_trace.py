import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from godotgaussiansplatting_amd import capi, scenes
cfg = sys.argv[1]
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
ctx = capi.Context(n, w, h)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
fr = capi.make_frame(vp, cam)
for k in range(int(sys.argv[2])):
    t0 = time.perf_counter(); ctx.render(fr); ctx.synchronize(); dt = time.perf_counter() - t0
    if k % int(sys.argv[3]) == 0:
        st = ctx.stats(); print(k, round(dt * 1e3, 3), st["pairs_round"], st["num_sorted"], st["num_composited"])
# then a long unpaced burst (the host runs ahead of the GPU): re-checks must stay short
t0 = time.perf_counter()
for k in range(1500):
    ctx.render(fr)
ctx.synchronize()
print("unpaced burst:", round(1500 / (time.perf_counter() - t0), 1), "fps", ctx.stats()["pairs_round"])
