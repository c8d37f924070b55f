#!/usr/bin/env python3
"""Stress: scenes whose splats cover many tiles (real captures have D/N ~ 10 and a few huge background splats)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from godotgaussiansplatting_amd import capi, scenes
n, w, h = 1_000_000, 1920, 1080
cam = scenes.default_camera()
vp, pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far)
fr = capi.make_frame(vp, pos)
for scale_n, huge in ((1_000_000, 0), (30_000, 0), (30_000, 2000), (3_000, 0)):
    rows = scenes.synthetic_rows(n, 9, 0, scale_n=scale_n)
    if huge:
        rows[:huge, 55:58] += np.log(30.0)      # a few thousand splats 30x larger (hundreds of tiles each)
    ctx = capi.Context(n, w, h, flags=capi.FLAG_TIMING | capi.FLAG_KERNEL_TIMING, key_budget_factor=60)
    ctx.upload_ply_rows(rows)
    for _ in range(3):
        ctx.render(fr)
    st = ctx.stats()
    print(f"scale_n={scale_n} huge={huge}: D/N={st['num_emitted']/n:.1f} overflow={st['overflow']} "
          f"proj {st['ms_projection']:.3f} sort {st['ms_sort']:.3f} bounds {st['ms_boundaries']:.3f} render {st['ms_render']:.3f} | "
          + " ".join(f"{k}={v:.3f}" for k, v in st['ms_kernel'].items() if k in ('project','scan','emit')))
    ctx.close()
