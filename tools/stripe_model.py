#!/usr/bin/env python3
"""Per-rank frame time of the tile-stripe shard, measured on ONE GPU: for G = 1,2,4,8 every rank's stripe is rendered
in turn by the same context and timed; max over ranks = the compute part of a G-GPU frame (the RCCL gather overlaps
it).  Tells how far the replicated part of the projection and the latency-bound sort limit strong scaling."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import bench
from godotgaussiansplatting_amd import capi
from godotgaussiansplatting_amd.distributed import even_cuts, balanced_cuts, time_balanced_cuts

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
MORTON = len(sys.argv) > 2 and sys.argv[2] in ("morton", "cull", "cull+ties")
CULL = len(sys.argv) > 2 and sys.argv[2] in ("cull", "cull+ties")   # + GSPLAT_FLAG_BLOCK_CULL, frames as render_begin / render_end
TIES = len(sys.argv) > 2 and sys.argv[2] == "cull+ties"             # + GSPLAT_FLAG_TIES_STORAGE_ORDER (no tie repair, 16-bit keys)
FLAGS = (capi.FLAG_BLOCK_CULL if CULL else 0) | (capi.FLAG_TIES_STORAGE_ORDER if TIES else 0)
ONLY_G = [int(x) for x in __import__("os").environ.get("STRIPE_MODEL_G", "1,2,4,8").split(",")]
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
from godotgaussiansplatting_amd import scenes
ROWS = scenes.config_rows(cfg)


def upload(c):
    for first in range(0, n, 1 << 20):
        c.upload_ply_rows(ROWS[first:first + (1 << 20)], first=first, load_time=-10.0)
    if MORTON:
        c.finalize_scene()   # gsplat_finalize_scene: Morton re-layout of the stored scene


ctx = capi.Context(n, w, h, flags=capi.FLAG_TIMING | FLAGS)
upload(ctx)
fr = capi.make_frame(vp, cam)
gx, gy = (w + 15) // 16, (h + 15) // 16
ctx.render(fr); ctx.synchronize()
if CULL:
    import torch
    TOP = torch.zeros(1, dtype=torch.int32, device="cuda")   # the frame's highest populated tile + 1 (what the
    ctx.render_begin(fr, TOP.data_ptr()); ctx.render_end(); ctx.synchronize()   # all-reduce(MAX) would deliver)


def render(c):
    if CULL:
        c.render_begin(fr)
        c.render_end(frame_last_tile_ptr=TOP.data_ptr())
    else:
        c.render(fr)


b = ctx.read_bounds().astype(np.int64)
cols = np.clip(b[:, 1] - b[:, 0], 0, None).reshape(gy, gx).sum(0).astype(float)


def time_stripe(b0, b1, reps=30):
    ctx.set_stripe(capi.STRIPE_COLUMNS, b0, b1)
    for _ in range(3):
        render(ctx)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        render(ctx)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    st = ctx.stats()
    return dt, st


out = {}
for G in ONLY_G:
    for name, cuts in (("even", even_cuts(gx, G)), ("balanced", balanced_cuts(cols + 64.0 * gy, G))):
        if G == 1 and name == "balanced":
            continue
        ts, rows = [], []
        for r in range(G):
            dt, st = time_stripe(cuts[r], cuts[r + 1])
            ts.append(dt)
            rows.append((round(dt, 3), st["num_sorted"], round(st["ms_projection"], 3), round(st["ms_sort"], 3), round(st["ms_render"], 3)))
        out[f"G{G}_{name}"] = {"max_ms": max(ts), "mean_ms": float(np.mean(ts)), "fps_bound": 1e3 / max(ts), "ranks": rows}
        print(f"G={G} {name}: max {max(ts):.3f} ms  mean {np.mean(ts):.3f} ms  -> <= {1e3/max(ts):.0f} fps   {rows if G<=4 else rows[:4]}")
# ... and re-cut from the MEASURED time of every rank (what bench.py --gpus N does after its pairs-based first guess):
# three rounds of distributed.time_balanced_cuts
TIMED = {}
for G in [g for g in ONLY_G if g >= 4]:
    prior = cols + 64.0 * gy
    cuts = balanced_cuts(prior, G)
    for it in range(3):
        ts = [time_stripe(cuts[r], cuts[r + 1], reps=20)[0] for r in range(G)]
        cuts = time_balanced_cuts(cuts, ts, prior=prior)
    ts = [time_stripe(cuts[r], cuts[r + 1])[0] for r in range(G)]
    TIMED[G] = cuts
    out[f"G{G}_time_balanced"] = {"max_ms": max(ts), "mean_ms": float(np.mean(ts)), "cuts": cuts, "ranks_ms": [round(t, 3) for t in ts]}
    print(f"G={G} time-balanced (3 rounds): max {max(ts):.3f} ms  mean {np.mean(ts):.3f} ms  -> <= {1e3/max(ts):.0f} fps   cuts {cuts}  {[round(t, 3) for t in ts]}")
json.dump(out, open("gpurun_out/stripe_model_%s.json" % cfg, "w"), indent=1)

# frames in flight per rank: R contexts (views of the one scene) render the SAME stripe concurrently on their own
# streams; per-frame time per rank
ctx.set_stripe(capi.STRIPE_NONE, 0, 0)
for G, r in [(g, rr) for g in (4, 8) if g in ONLY_G for rr in ((g // 2 - 1, 0) if g == 8 else (g // 2 - 1,))]:
    cuts = TIMED.get(G) or balanced_cuts(cols + 64.0 * gy, G)   # (the time-balanced cuts: every rank about equally slow)
    for R in [int(x) for x in __import__("os").environ.get("STRIPE_FRAMES_IN_FLIGHT", "1,2,3,4").split(",")]:
        ring = [ctx.view(stripe=(capi.STRIPE_COLUMNS, cuts[r], cuts[r + 1]), flags=FLAGS) for _ in range(R)]
        for k in range(3 * R):
            render(ring[k % R])
        for c in ring:
            c.synchronize()
        reps = 60
        t0 = time.perf_counter()
        for k in range(reps):
            render(ring[k % R])
        for c in ring:
            c.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        print(f"G={G} rank {r} stripe, {R} frame(s) in flight: {dt:.3f} ms/frame -> {1e3/dt:.0f} fps per rank-equivalent")
        for c in ring:
            c.close()
ctx.close()
