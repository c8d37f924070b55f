#!/usr/bin/env python3
"""CPU only.  Round 4's review, item 5: the compositor's exponent y = hx dx^2 + hy dx dy + hz dy^2 of four splats x 64
pixels on the matrix pipe needs the EXPANDED form about a fixed origin, y = c0 + c1 u + c2 v + c3 u^2 + c4 uv + c5 v^2 (u,
v = the pixel's offset from the origin: small integers, exact), accumulated as a chain of six binary32 FMAs (what
v_mfma_f32_4x4x1 does).  The centred form of the contract never subtracts large numbers; the expanded one does (c0 = the
exponent AT the origin, up to ~10^3 in magnitude for a sharp splat 10-20 px away).  This script bounds that cancellation
on the records a frame really stages: for the (tile, splat) pairs of sampled tiles of a configuration, over the pixels
where the splat is visible (y >= -32), the error of the expanded chain and of the contract's own chain against binary64,
with the origin at the tile's corner (one coefficient set per staged splat) and at each 8x8 quadrant's corner (four).
usage: python tools/mfma_expanded_error.py [config = c3] [splats = 600000] [tiles = 400]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from godotgaussiansplatting_amd import capi, scenes  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600_000
ntiles = int(sys.argv[3]) if len(sys.argv) > 3 else 400
_, deg, w, h, seed = scenes.CONFIGS[cfg]
rows = scenes.config_rows(cfg, n)
rec = oracle.records_from_ply_rows(rows, -10.0)
cam = scenes.default_camera()
vp, cam_pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far)
out = oracle.render_frame(rec, oracle.Frame.make(vp, cam_pos, w, h), capacity=60 * n, want_image=False)
culled, values, bounds = out["culled"], out["values"], out["bounds"].astype(np.int64)
gx = (w + 15) // 16
LOG2E = np.float32(1.4426950408889634)
f32 = np.float32


def fma32(a, b, c):   # binary32 fma: the double product of two floats is exact; one rounding of the sum (double rounding
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)   # cases: ~2^-29 of inputs)


rng = np.random.default_rng(1)
lens = np.clip(bounds[:, 1] - bounds[:, 0], 0, None)
tiles = rng.choice(np.flatnonzero(lens > 0), size=min(ntiles, int((lens > 0).sum())), replace=False)
uu, vv = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32))
worst = {"contract": 0.0, "tile origin": 0.0, "quadrant origin": 0.0}
hist = {k: [] for k in worst}
pairs = 0
for t in tiles:
    ids = values[bounds[t, 0]:bounds[t, 1]][:256]          # (the first batch: what every tile stages)
    r = culled[ids]
    ipx, ipy = r[:, 0], r[:, 1]
    hx = (f32(-0.5) * r[:, 4]) * LOG2E
    hy = (-r[:, 5]) * LOG2E
    hz = (f32(-0.5) * r[:, 6]) * LOG2E
    ox, oy = f32((t % gx) * 16), f32((t // gx) * 16)
    px, py = (ox + uu).ravel()[None, :], (oy + vv).ravel()[None, :]           # (1, 256)
    dx64 = ipx[:, None].astype(np.float64) - px
    dy64 = ipy[:, None].astype(np.float64) - py
    y_true = hx[:, None].astype(np.float64) * dx64 * dx64 + hy[:, None].astype(np.float64) * dx64 * dy64 + hz[:, None].astype(np.float64) * dy64 * dy64
    vis = y_true >= -32.0
    if not vis.any():
        continue
    pairs += ids.size
    # the contract's chain
    dx, dy = (ipx[:, None] - px).astype(np.float32), (ipy[:, None] - py).astype(np.float32)
    a1 = (hx[:, None] * dx).astype(np.float32)
    a1 = fma32(np.broadcast_to(hy[:, None], dy.shape), dy, a1)
    yc = ((hz[:, None] * dy).astype(np.float32) * dy).astype(np.float32)
    yc = fma32(a1, dx, yc)
    e = np.abs(yc.astype(np.float64) - y_true)[vis]
    hist["contract"].append(e)

    def expanded(x0, y0, u, v):
        X = ipx.astype(np.float64) - x0
        Y = ipy.astype(np.float64) - y0
        H = [hx.astype(np.float64), hy.astype(np.float64), hz.astype(np.float64)]
        c = [H[0] * X * X + H[1] * X * Y + H[2] * Y * Y, -2 * H[0] * X - H[1] * Y, -H[1] * X - 2 * H[2] * Y, H[0], H[1], H[2]]
        c = [ci.astype(np.float32)[:, None] for ci in c]       # (coefficients rounded once from binary64: the best case)
        mono = [np.ones_like(u), u, v, u * u, u * v, v * v]
        acc = (c[0] * mono[0]).astype(np.float32)
        for k in range(1, 6):
            acc = fma32(np.broadcast_to(c[k], (ids.size, u.size)), np.broadcast_to(mono[k][None, :], (ids.size, u.size)), acc)
        return acc

    ye = expanded(np.float64(ox), np.float64(oy), uu.ravel(), vv.ravel())
    hist["tile origin"].append(np.abs(ye.astype(np.float64) - y_true)[vis])
    yq = np.empty_like(ye)
    for q in range(4):
        qx, qy = (q & 1) * 8, (q >> 1) * 8
        sel = ((uu.ravel() >= qx) & (uu.ravel() < qx + 8) & (vv.ravel() >= qy) & (vv.ravel() < qy + 8))
        yq[:, sel] = expanded(np.float64(ox) + qx, np.float64(oy) + qy, uu.ravel()[sel] - qx, vv.ravel()[sel] - qy)
    hist["quadrant origin"].append(np.abs(yq.astype(np.float64) - y_true)[vis])
print(f"{cfg}: {n:,} splats, {len(tiles)} tiles, {pairs:,} staged (tile, splat) pairs, pixels with y >= -32 only")
print("| form | max |dy| | p99.9 |dy| | relative error of alpha at the max (|dy| ln 2) |\n|---|---|---|---|")
for k, v in hist.items():
    e = np.concatenate(v)
    print(f"| {k} | {e.max():.3g} | {np.percentile(e, 99.9):.3g} | {e.max() * np.log(2):.3g} |")
