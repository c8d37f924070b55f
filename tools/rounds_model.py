#!/usr/bin/env python3
"""How many pairs would a two-round frame emit?  Round A = the front fraction f of the depth-sorted visible splats,
composited with the reference's batch/termination rules; round B = the remaining splats, emitting (all their pairs)
only if some tile of their rectangle is not finished after round A.  Uses one rendered frame's taps (tile_bounds,
staged count per tile, sorted pairs).  usage: rounds_model.py <config>"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3d"
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
ctx = capi.Context(n, w, h)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
fr = capi.make_frame(vp, cam)
ctx.render(fr); ctx.synchronize()
bounds = ctx.read_bounds().astype(np.int64)
staged = ctx.read_tile_staged().astype(np.int64)
keys, values = ctx.read_sorted()
ctx.close()
D = keys.size
tile = (keys >> 16).astype(np.int64)
depth = (keys & 0xFFFF).astype(np.uint64)
nt = np.clip(bounds[:, 1] - bounds[:, 0], 0, None)
T = nt.size
# depth rank of every visible splat
ids, first_idx, cnt = np.unique(values, return_index=True, return_counts=True)
skey = (depth[first_idx] << np.uint64(32)) | ids.astype(np.uint64)
order = np.argsort(skey, kind="stable")
rank_of = np.zeros(n, dtype=np.int64)
rank_of[ids[order]] = np.arange(ids.size)
V = ids.size
pairs_of = np.zeros(n, dtype=np.int64); pairs_of[ids] = cnt
prank = rank_of[values]
exits = staged < nt          # the tile left its loop at a batch boundary before the end of its list
print(f"{cfg}: N {n} V {V} D {D} tiles {T}  D_c {staged.sum()}  tiles that exit early {exits.sum()}")
for f in (0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.7):
    isA = prank < int(f * V)
    nA = np.bincount(tile[isA], minlength=T)
    done = exits & (staged <= nA)
    pairs_A = int(isA.sum())
    # B splats with at least one unfinished tile
    inB = ~isA
    need = np.bincount(values[inB], weights=(~done[tile[inB]]).astype(np.float64), minlength=n) > 0
    pairs_B = int(pairs_of[need].sum())
    splats_B = int(need.sum())
    print(f"  f={f:4.2f}: A pairs {pairs_A/1e6:7.2f} M   tiles done after A {done.sum():6d}/{T}   B splats {splats_B/1e6:6.2f} M  B pairs {pairs_B/1e6:7.2f} M"
          f"   total {(pairs_A+pairs_B)/1e6:7.2f} M = {100.0*(pairs_A+pairs_B)/D:5.1f} % of D")
