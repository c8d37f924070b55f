#!/usr/bin/env python3
"""Per-frame time of ONE rank's stripe of an 8-GPU shard when its frames go through the launch sequence in BATCHES
(gsplat_render_batch_begin / _end on a batch context), measured on one GPU:

    python tools/stripe_batch.py <config> [--batch 1,2,3,4] [--in-flight 1,2,3] [--ranks slowest,middle,edge] [--axis rows]

The scene as bench.py --gpus N renders it (Morton layout, GSPLAT_FLAG_BLOCK_CULL | GSPLAT_FLAG_TIES_STORAGE_ORDER), 8 stripes
cut like the bench cuts them (pairs-balanced, then three rounds of distributed.time_balanced_cuts from measured rank times with
plain frames); then, for each chosen rank, R contexts (views of the one scene) each keep a batch of B frames in flight on their
own streams: ms per FRAME = elapsed / (batches x B).  B = 1 is today's path (plain contexts, gsplat_render_begin / _end).
The frames of a batch are B DIFFERENT cameras (an orbit of 1 degree per frame around the bench's camera), as consecutive
frames of an animation are; `--fixed` repeats the bench's fixed camera instead."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes
from godotgaussiansplatting_amd.distributed import balanced_cuts, time_balanced_cuts

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--batch", default="1,2,3,4")
ap.add_argument("--in-flight", default="1,2,3,4")
ap.add_argument("--ranks", default="slowest,middle,edge")
ap.add_argument("--axis", default="rows", choices=["rows", "columns"])
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--fixed", action="store_true")
ap.add_argument("--reps", type=int, default=48)
ap.add_argument("--cuts-from-batch", type=int, default=0,
                help="re-cut the stripes from rank times measured with batches of this many frames (what bench.py --batch B does "
                     "since round 6) instead of plain frames")
args = ap.parse_args()
cfg, G = args.config, args.world
n, deg, w, h, seed, vp, cam_pos = bench.build_scene_inputs(cfg)
rows_all = scenes.config_rows(cfg)
FLAGS = capi.FLAG_BLOCK_CULL | capi.FLAG_TIES_STORAGE_ORDER
AX = capi.STRIPE_ROWS if args.axis == "rows" else capi.STRIPE_COLUMNS
gx, gy = (w + 15) // 16, (h + 15) // 16
units = gy if args.axis == "rows" else gx

# (no torch in this process: HIP maps streams onto its four hardware queues round-robin in creation order, and a second
# runtime user's streams shift which of the ring's contexts share a queue — call 3 of round 6 measured "2 in flight = 1 in
# flight" for exactly that reason.  The few device words this tool needs come from hipMalloc through ctypes.)
from godotgaussiansplatting_amd import _lib as _gl
_gl.load()                       # (loads the one HIP runtime of the process, globally)
_hip = ctypes.CDLL(None)         # ... whose symbols are then in the global scope


class DeviceWords:
    def __init__(self, count):
        self.count = count
        self.p = ctypes.c_void_p()
        assert _hip.hipMalloc(ctypes.byref(self.p), ctypes.c_size_t(4 * count)) == 0
        self.fill(0)

    def data_ptr(self):
        return self.p.value

    def fill(self, value):
        host = (ctypes.c_uint32 * self.count)(*([int(value)] * self.count))
        assert _hip.hipMemcpy(self.p, host, ctypes.c_size_t(4 * self.count), 1) == 0

    def item(self, i=0):
        host = (ctypes.c_uint32 * self.count)()
        assert _hip.hipDeviceSynchronize() == 0
        assert _hip.hipMemcpy(host, self.p, ctypes.c_size_t(4 * self.count), 2) == 0
        return int(host[i])


owner = capi.Context(n, w, h, flags=capi.FLAG_TIMING | FLAGS)
for first in range(0, n, 1 << 20):
    owner.upload_ply_rows(rows_all[first:first + (1 << 20)], first=first, load_time=-10.0)
owner.finalize_scene()
fixed = capi.make_frame(vp, cam_pos)
owner.render(fixed); owner.synchronize()
top1 = DeviceWords(1)
owner.render_begin(fixed, top1.data_ptr()); owner.render_end(); owner.synchronize()
b = owner.read_bounds().astype(np.int64)
cnt = np.clip(b[:, 1] - b[:, 0], 0, None).reshape(gy, gx)
prior = (cnt.sum(axis=1) if args.axis == "rows" else cnt.sum(axis=0)).astype(float) + 64.0 * (gx if args.axis == "rows" else gy)


def cams(count):
    if args.fixed:
        return [fixed] * count
    base = scenes.default_camera()
    r = float(np.linalg.norm(base.origin))
    out = []
    for k in range(count):
        a = np.deg2rad(1.0 * k)
        c = scenes.look_at_camera((r * np.sin(a), base.origin[1], r * np.cos(a)))
        v, p = capi.make_view_proj(c.xform12(), c.fov, w / h, c.near, c.far)
        out.append(capi.make_frame(v, p))
    return out


def time_plain(c, b0, b1, reps=20):
    c.set_stripe(AX, b0, b1)
    for _ in range(3):
        c.render_begin(fixed); c.render_end(frame_last_tile_ptr=top1.data_ptr())
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.render_begin(fixed); c.render_end(frame_last_tile_ptr=top1.data_ptr())
    c.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def time_batched(c, b0, b1, B, reps=8):
    c.set_stripe(AX, b0, b1)
    fl = [fixed] * B
    for _ in range(2):
        c.render_batch_begin(fl); c.render_batch_end(topB.data_ptr())
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.render_batch_begin(fl); c.render_batch_end(topB.data_ptr())
    c.synchronize()
    return (time.perf_counter() - t0) / (reps * B) * 1e3


cuts = balanced_cuts(prior, G)
if args.cuts_from_batch > 1:
    topB = DeviceWords(4)
    topB.fill(top1.item())
    bview = owner.view(flags=FLAGS, batch=args.cuts_from_batch)
for _ in range(3):
    if args.cuts_from_batch > 1:
        ts = [time_batched(bview, cuts[r], cuts[r + 1], args.cuts_from_batch) for r in range(G)]
    else:
        ts = [time_plain(owner, cuts[r], cuts[r + 1]) for r in range(G)]
    cuts = time_balanced_cuts(cuts, ts, prior=prior)
if args.cuts_from_batch > 1:
    print(f"{cfg} cuts re-cut from batches of {args.cuts_from_batch}: {cuts}; batched ms per frame per rank {[round(t, 3) for t in ts]}", flush=True)
    bview.close()
ts = [time_plain(owner, cuts[r], cuts[r + 1], reps=30) for r in range(G)]
print(f"{cfg} G={G} {args.axis} time-balanced cuts {cuts} plain one-at-a-time ms per rank {[round(t, 3) for t in ts]}", flush=True)
owner.set_stripe(capi.STRIPE_NONE, 0, 0)
which = {"slowest": int(np.argmax(ts)), "middle": G // 2 - 1, "edge": 0}
out = {"config": cfg, "world": G, "axis": args.axis, "cuts": [int(c) for c in cuts], "plain_ms": ts, "moving_camera": not args.fixed, "ranks": {}}
for label in args.ranks.split(","):
    r = which[label]
    stripe = (AX, cuts[r], cuts[r + 1])
    res = {}
    for B in [int(x) for x in args.batch.split(",")]:
        frames = cams(B)
        tops = DeviceWords(4)
        tops.fill(top1.item())
        for R in [int(x) for x in args.in_flight.split(",")]:
            if B == 1:
                ring = [owner.view(stripe=stripe, flags=FLAGS) for _ in range(R)]

                def go(c, k):
                    c.render_begin(frames[0]); c.render_end(frame_last_tile_ptr=tops.data_ptr())
            else:
                ring = [owner.view(stripe=stripe, flags=FLAGS, batch=B) for _ in range(R)]

                def go(c, k):
                    c.render_batch_begin(frames); c.render_batch_end(tops.data_ptr())
            for k in range(4 * R):
                go(ring[k % R], k)
            for c in ring:
                c.synchronize()
            reps = max(R * 4, args.reps // B)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for k in range(reps):
                    go(ring[k % R], k)
                for c in ring:
                    c.synchronize()
                best = min(best, (time.perf_counter() - t0) / (reps * B) * 1e3)
            st = ring[0].stats()
            res[f"B{B}_R{R}"] = best
            print(f"  rank {r} ({label}) batch {B} x {R} in flight: {best:.4f} ms per frame -> {1e3 / best:.0f} fps"
                  f"   (pairs per launch sequence {st['num_sorted']}, sort passes {st['sort_passes']})", flush=True)
            for c in ring:
                c.close()
    # per kernel class, one launch sequence at a time (HIP events between the launches): ms per FRAME
    kern = {}
    for B in [int(x) for x in args.batch.split(",")]:
        frames = cams(B)
        tops = DeviceWords(4)
        tops.fill(top1.item())
        c = owner.view(stripe=stripe, flags=FLAGS, batch=B) if B > 1 else owner.view(stripe=stripe, flags=FLAGS)
        c.set_timing(capi.FLAG_TIMING | capi.FLAG_KERNEL_TIMING)
        acc = []
        for k in range(14):
            if B > 1:
                c.render_batch_begin(frames); c.render_batch_end(tops.data_ptr())
            else:
                c.render_begin(frames[0]); c.render_end(frame_last_tile_ptr=tops.data_ptr())
            st = c.stats()
            if k >= 4:
                acc.append([st["ms_kernel"][q] for q in st["ms_kernel"]] + [st["ms_total"]])
        m = np.median(np.array(acc), axis=0) / B
        kern[f"B{B}"] = {q: round(float(v), 4) for q, v in zip(list(st["ms_kernel"].keys()) + ["total"], m)}
        print(f"  rank {r} ({label}) batch {B}, per frame by kernel class: {kern[f'B{B}']}", flush=True)
        c.close()
    out["ranks"][label] = {"rank": r, "ms_per_frame": res, "best": min(res.values()), "best_setting": min(res, key=res.get),
                           "ms_per_frame_by_kernel_class": kern}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/stripe_batch_{cfg}.json", "w"), indent=1)
owner.close()
