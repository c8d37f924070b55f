#!/usr/bin/env python3
"""CPU only: pairs per splat (D/N), visible fraction and the per-tile pair distribution of the capture-like scene
(scenes.capture_like_rows, config c3r) as a function of its median splat size, from the oracle on a thinned sample at
1080p with the default camera — how scenes.CAPTURE_SIZE was chosen.  usage: python tools/calibrate_c3r.py [n] [size ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from godotgaussiansplatting_amd import capi, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
sizes = [float(x) for x in sys.argv[2:]] or [scenes.CAPTURE_SIZE]
full, deg, w, h, seed = scenes.CONFIGS["c3r"]
cam = scenes.default_camera()
vp, cam_pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far) if os.path.exists(
    os.path.join(os.path.dirname(os.path.abspath(capi.__file__)), "libgsplat_hip.so")) else (None, None)
for size in sizes:
    # a thinned sample keeps D/N only if its splats keep their size: D scales with n, the per-tile lists with n / full
    rows = scenes.capture_like_rows(n, seed, deg, size=size)
    rec = oracle.records_from_ply_rows(rows, -10.0)
    fr = oracle.Frame.make(vp, cam_pos, w, h)
    out = oracle.render_frame(rec, fr, capacity=60 * n, want_image=False)
    b = out["bounds"].astype(np.int64)
    per_tile = np.clip(b[:, 1] - b[:, 0], 0, None) * (full / n)
    cnt = out["counts"]
    print(f"size {size}: D/N {out['D'] / n:.2f}  visible {out['stats']['visible'] / n:.3f}  pairs/tile at full N: median "
          f"{np.median(per_tile):.0f} p90 {np.percentile(per_tile, 90):.0f} p99 {np.percentile(per_tile, 99):.0f} max {per_tile.max():.0f}"
          f"  splats over 512 tiles: {(cnt > 512).sum()} largest {cnt.max()}")
