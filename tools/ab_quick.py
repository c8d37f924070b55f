#!/usr/bin/env python3
"""One short line per (library, config): the numbers an A/B between two builds of libgsplat_hip.so is decided on.

    GSPLAT_LIB=build_variants/libgsplat_x.so python tools/ab_quick.py c3 [steps]

Prints JSON: frames/s one at a time and with two in flight, GPU frame time (HIP events, first to last kernel) and the
per-kernel-class milliseconds / launch counts of a frame.  Same scene, camera and settle phase as bench.py; run the
variants back to back inside ONE gpurun call (box-to-box spread is a few percent).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from godotgaussiansplatting_amd import capi, scenes  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n, deg, w, h, seed = scenes.CONFIGS[cfg]
    cam = scenes.default_camera()
    vp, cam_pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far)
    frame = capi.make_frame(vp, cam_pos)
    rows = scenes.config_rows(cfg)
    ctx = capi.Context(n, w, h)
    for first in range(0, n, 1 << 20):
        ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
    if os.environ.get("AB_FINALIZE") == "1":
        ctx.finalize_scene()
    ring = [ctx, ctx.view()]
    for c in ring:
        for _ in range(int(os.environ.get("AB_SETTLE", "160"))):
            c.render(frame)
            c.synchronize()
    out = {"lib": os.path.basename(os.environ.get("GSPLAT_LIB", "libgsplat_hip.so")), "config": cfg,
           "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GSPLAT_") and k not in ("GSPLAT_LIB", "GSPLAT_COMMIT")}}
    best_seq, best_two = 0.0, 0.0
    for _ in range(3):  # best of three short legs each: the decision is between builds, not between moments
        for _ in range(20):
            ctx.render(frame)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.render(frame)
        ctx.synchronize()
        best_seq = max(best_seq, steps / (time.perf_counter() - t0))
        for k in range(20):
            ring[k % 2].render(frame)
        for c in ring:
            c.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            ring[k % 2].render(frame)
        for c in ring:
            c.synchronize()
        best_two = max(best_two, steps / (time.perf_counter() - t0))
    out["fps_one_at_a_time"] = round(best_seq, 1)
    out["fps_two_in_flight"] = round(best_two, 1)
    ctx.set_timing(capi.FLAG_TIMING)
    tot = []
    for _ in range(40):
        ctx.render(frame)
        st = ctx.stats()
        tot.append([st["ms_projection"], st["ms_sort"], st["ms_boundaries"], st["ms_render"], st["ms_total"]])
    m = np.median(np.array(tot), axis=0)
    out["ms_pass"] = {k: round(float(v), 4) for k, v in zip(["projection", "sort", "boundaries", "render", "total"], m)}
    ctx.set_timing(capi.FLAG_TIMING | capi.FLAG_KERNEL_TIMING)
    rows_k = []
    for _ in range(40):
        ctx.render(frame)
        st = ctx.stats()
        rows_k.append([st["ms_kernel"][k] for k in st["ms_kernel"]])
    km = np.median(np.array(rows_k), axis=0)
    out["ms_kernel"] = {k: round(float(v), 4) for k, v in zip(st["ms_kernel"].keys(), km)}
    out["launches"] = {k: int(v) for k, v in st["launches_kernel"].items()}
    out["pairs_round"] = st["pairs_round"]
    out["D"] = st["num_sorted"]
    out["lazy"] = st["lazy_colors"]
    ctx.set_timing(0)
    print(json.dumps(out), flush=True)
    for c in reversed(ring):
        c.close()


if __name__ == "__main__":
    main()
