#!/usr/bin/env python3
"""Sort pass time vs pair count for the two partition sizes (GSPLAT_SORT_SMALL=0 forces 4096-key partitions):
column stripes of growing width on the c3 scene give D from ~0.1 M to 10 M."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes

n, deg, w, h, seed, vp, cam = bench.build_scene_inputs("c3")
rows = scenes.synthetic_rows(n, seed, 0)
fr = capi.make_frame(vp, cam)
gx = (w + 15) // 16
res = {}
for mode in ("0", "4000000"):  # the second value is capped at the built-in threshold: run with a library built with a larger SMALL_COUNT to see beyond it
    os.environ["GSPLAT_SORT_SMALL"] = mode   # capped at the built-in default inside the library
    ctx = capi.Context(n, w, h, flags=capi.FLAG_TIMING, sh_degree=0)
    for first in range(0, n, 1 << 20):
        ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first)
    for width in (2, 4, 6, 8, 10, 12, 16, 24, 40, 120):
        b = gx // 2 - width // 2
        ctx.set_stripe(capi.STRIPE_COLUMNS, b, b + width)
        ts = []
        for k in range(25):
            ctx.render(fr)
            st = ctx.stats()
            if k >= 5:
                ts.append(st["ms_sort"])
        res.setdefault(width, {})[mode] = (st["num_sorted"], float(np.median(ts)))
    ctx.close()
for width, r in res.items():
    d = r["0"][0]
    print(f"width {width:4d}  D={d/1e6:6.2f} M   4096-key {r['0'][1]:.4f} ms   1024-key {r['4000000'][1]:.4f} ms")
