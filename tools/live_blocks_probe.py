#!/usr/bin/env python3
"""How evenly do the projection workgroups a stripe rank does NOT skip spread over the XCDs?  project_kernel and pass 0 of
the splat sort hand XCD x the contiguous eighth x of the 512-slot blocks (write locality); with block culling two thirds of
a stripe rank's blocks leave at once, and the live ones cluster along the Morton curve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
ctx = capi.Context(n, w, h, flags=capi.FLAG_BLOCK_CULL | capi.FLAG_TIES_STORAGE_ORDER)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
ctx.finalize_scene()
fr = capi.make_frame(vp, cam)
gy = (h + 15) // 16
for (a, b) in ((0, 17), (29, 34), (34, 38), (51, 68)) if cfg == "c3" else ((0, 29), (57, 68), (107, 135)):
    ctx.set_stripe(capi.STRIPE_ROWS, a, b)
    ctx.render_begin(fr); ctx.render_end(); ctx.synchronize()
    bs = ctx.read_block_sums()
    live = bs[:, 3] == 0
    nb = live.size
    per = (nb + 7) // 8
    eighths = [int(live[x * per:(x + 1) * per].sum()) for x in range(8)]
    vis = [int(bs[x * per:(x + 1) * per, 1].sum()) for x in range(8)]
    print(f"{cfg} rows {a}:{b}: live blocks {int(live.sum())} of {nb}; per contiguous eighth (= per XCD): {eighths}  max/mean {max(eighths) / (sum(eighths) / 8):.2f}; visible splats per eighth {vis} max/mean {max(vis) / (sum(vis) / 8 + 1e-9):.2f}")
ctx.close()
