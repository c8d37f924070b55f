#!/usr/bin/env python3
"""Per-kernel-class times of ONE rank's stripe of a G-GPU shard (Morton layout + block culling), on one GPU.
usage: python tools/stripe_kernels.py <config> <G> <rank> [ties]   (ties: GSPLAT_FLAG_TIES_STORAGE_ORDER; GSPLAT_PAIR_SORT=split|wide
selects the pair-level sort form)"""
import os
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes
from godotgaussiansplatting_amd.distributed import balanced_cuts

cfg, G, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
TIES = capi.FLAG_TIES_STORAGE_ORDER if len(sys.argv) > 4 and sys.argv[4] == "ties" else 0
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
ctx = capi.Context(n, w, h, flags=capi.FLAG_BLOCK_CULL | TIES)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
ctx.finalize_scene()
fr = capi.make_frame(vp, cam)
gx, gy = (w + 15) // 16, (h + 15) // 16
ctx.render(fr)
b = ctx.read_bounds().astype(np.int64)
cols = np.clip(b[:, 1] - b[:, 0], 0, None).reshape(gy, gx).sum(0).astype(float)
cuts = balanced_cuts(cols + 64.0 * gy, G)
import torch
top = torch.zeros(1, dtype=torch.int32, device="cuda")
ctx.render_begin(fr, top.data_ptr()); ctx.render_end(); ctx.synchronize()
ctx.set_stripe(capi.STRIPE_COLUMNS, cuts[rank], cuts[rank + 1])
ctx.set_timing(capi.FLAG_TIMING | capi.FLAG_KERNEL_TIMING)
acc = []
for _ in range(30):
    ctx.render_begin(fr)
    ctx.render_end(frame_last_tile_ptr=top.data_ptr())
    st = ctx.stats()
    acc.append([st["ms_kernel"][k] for k in st["ms_kernel"]] + [st["ms_total"]])
m = np.median(np.array(acc), axis=0)
print(cfg, "G", G, "rank", rank, "cuts", cuts[rank], cuts[rank + 1], "D", st["num_sorted"], "V", st["num_visible"],
      "sort_passes", st["sort_passes"], "key bytes", st["pair_key_bytes"], "ties" if TIES else "", os.environ.get("GSPLAT_PAIR_SORT", ""))
print({k: round(float(v), 4) for k, v in zip(list(st["ms_kernel"].keys()) + ["total"], m)}, st["launches_kernel"])
skipped = ctx.read_block_sums()[:, 3].sum()
print("blocks skipped", int(skipped), "of", (n + 511) // 512)
