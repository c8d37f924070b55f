#!/usr/bin/env python3
"""bench.py — frames/s of the forward Gaussian-splat hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3]

A "step" is one frame: projection -> key sort -> tile ranges -> compositor over one synthetic scene that is
already resident in HBM (SURVEY.md §8d synthetic generator, fixed camera).  N=1 runs one context on cuda:0.
N>1 is launched by torch.distributed.run, one rank per GPU: the frame is sharded by tile-column stripes and
the finished stripes are all-gathered with RCCL every frame (strong scaling: the frame is fixed, the work is
split).  Rank 0 prints ONE JSON line.

Extra objects on the line (N=1): "roofline" for the dominant kernel (algorithmic bytes per launch / average launch
time measured with HIP events on the context's stream) and "cpu_baseline" (the CPU oracle — a port of the
reference pipeline, not the Godot/Vulkan path, which cannot run here — on a bounded sample, timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from godotgaussiansplatting_amd import capi, scenes  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_scene_inputs(cfg_name):
    n, deg, w, h, seed = scenes.CONFIGS[cfg_name]
    cam = scenes.default_camera()
    vp, cam_pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far)
    return n, deg, w, h, seed, vp, cam_pos


FINALIZE = [False]  # gsplat_finalize_scene after the upload (set in main)


_ROWS = {}  # the synthetic .ply rows, generated once per process (every context of the ring loads the same scene)


CONFIG_NAME = ["c3"]


def upload_scene(ctx, n, seed, deg, chunk=1 << 20):
    key = CONFIG_NAME[0]
    if key not in _ROWS:
        _ROWS.clear()
        _ROWS[key] = scenes.config_rows(key)
    rows = _ROWS[key]
    for first in range(0, n, chunk):
        ctx.upload_ply_rows(rows[first:first + chunk], first=first, load_time=-10.0)
    if FINALIZE[0]:
        ctx.finalize_scene()
    return rows


def kernel_algorithmic_bytes(st):
    """SURVEY.md §8(d) per-frame algorithmic bytes, split per kernel class (per launch for the sort passes)."""
    N, V, D, Dc = st["num_splats"], st["num_visible"], st["num_sorted"], st["num_composited"]
    K = (st["sh_degree"] + 1) ** 2
    T = st["_tiles"]
    P = st["_pixels"]
    passes = st["sort_passes"]
    # SURVEY.md §8(d) counts the SH coefficients (12 K bytes per visible splat) in the projection pass; this build lets
    # the compositor read them instead, only for the pairs it stages, in frames where that is cheaper:
    # the bytes move with the work
    lazy = bool(st.get("lazy_colors"))
    return {
        "project": 16 * N + 28 * V + 48 * V + (0 if lazy else 12 * K * V),
        "emit": 8 * D,
        "sort_upsweep": 4 * D / passes,          # the one key read for histograms, spread over the passes
        "sort_downsweep": 16 * D,                # per launch: read + write 8 B pairs
        "boundaries": 4 * D + 8 * T,
        "render": (40 + (12 * K if lazy else 0)) * Dc + 16 * P,
        "tile_sort": 16 * D,                     # per-tile depth sort: every pair read and written once
    }


def phase_algorithmic_bytes(st):
    N, V, D, Dc = st["num_splats"], st["num_visible"], st["num_sorted"], st["num_composited"]
    K = (st["sh_degree"] + 1) ** 2
    return {"projection": 16 * N + (28 + 12 * K) * V + 48 * V + 8 * D, "sort": 4 * D + st["sort_passes"] * 16 * D,
            "boundaries": 4 * D + 8 * st["_tiles"], "render": 40 * Dc + 16 * st["_pixels"]}


def cpu_baseline(cfg_name, vp, cam_pos, budget_splats=8_000_000):
    """The oracle (a CPU port of the reference's four passes) timed on this host's cores on a bounded sample of
    the same workload: same camera, resolution and splat-size law, first min(N, budget) splats of the scene."""
    import oracle
    n, deg, w, h, seed = scenes.CONFIGS[cfg_name]
    ns = min(n, budget_splats)
    rows = scenes.config_rows(cfg_name, ns)
    rec = oracle.records_from_ply_rows(rows, -10.0)
    fr = oracle.Frame.make(vp, cam_pos, w, h)
    oracle.render_frame(rec[: min(ns, 20000)], fr)  # warm the library / OpenMP pool
    t0 = time.perf_counter()
    frames = 0
    while True:
        out = oracle.render_frame(rec, fr)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or frames >= 5:
            break
    fps = frames / dt
    what = "the whole scene" if ns == n else (f"the first {ns:,} of {n:,} splats (same size law); value = measured "
                                              f"{fps:.3f} frames/s x {ns}/{n} (the passes are linear in N)")
    return {"value": fps * ns / n, "unit": "frames/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{frames} frame(s) of {what}, SH deg {deg}, {w}x{h}, D={out['D']:,}; CPU restatement of the "
                      f"reference pipeline (oracle/, OpenMP), not Godot/Vulkan",
            "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=os.environ.get("GSPLAT_BENCH_CONFIG", "c3"), choices=sorted(scenes.CONFIGS))
    ap.add_argument("--axis", default="columns", choices=["columns", "rows"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fast-exp", action="store_true", help="GSPLAT_FLAG_FAST_EXP (not the parity default)")
    ap.add_argument("--no-rebalance", action="store_true")
    ap.add_argument("--finalize", choices=["auto", "on", "off"], default="auto",
                    help="gsplat_finalize_scene (Morton re-layout of the stored scene) after loading; auto = only "
                         "for N>1, where it cuts the replicated part of the projection")
    ap.add_argument("--frames-in-flight", type=int, default=int(os.environ.get("GSPLAT_FRAMES_IN_FLIGHT", "2")),
                    help="N=1: frames kept in flight (FrameRing: one context = stream + intermediate buffers + scene "
                         "replica per slot; like RenderingDevice's frame queue).  Every frame runs the whole pipeline; "
                         "the HBM-bound projection of one frame overlaps the issue-bound compositing of the previous "
                         "one.  1 = strictly one frame at a time (also reported as sequential_fps).")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # convenience: `python bench.py --gpus N` re-launches itself the way the driver does (one rank per GPU)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # stdout carries exactly ONE JSON line.  Libraries loaded later write banners to the C-level stdout (RCCL prints
    # its version block there), so fd 1 is pointed at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GSPLAT_FORCE_DIST=1: take the multi-rank code path (RCCL process group, stripe contexts, pipelined all-gather)
    # even with one rank — the only way to exercise the RCCL calls on a single-GPU box
    force_dist = os.environ.get("GSPLAT_FORCE_DIST") == "1"
    multi = world > 1 or force_dist
    FINALIZE[0] = args.finalize == "on" or (args.finalize == "auto" and world > 1)
    CONFIG_NAME[0] = args.config
    n, deg, w, h, seed, vp, cam_pos = build_scene_inputs(args.config)
    frame = capi.make_frame(vp, cam_pos)
    flags = capi.FLAG_FAST_EXP if args.fast_exp else 0

    dist = None
    torch = None
    sequential_fps = None
    if multi:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29534"))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    # N>1: every context launches on its own torch stream, so RCCL (which orders itself against the stream that is
    # current when the collective is issued) needs no host synchronisation between stripe render and all-gather;
    # four contexts per rank keep four frames in flight, like FrameRing on one GPU (per-rank work at 4-8 GPUs is small and
    # latency-bound: tools/stripe_model.py c3 cull measures 0.33 / 0.21 / 0.25 / 0.21 ms per frame and rank of 8 with
    # 1 / 2 / 3 / 4 in flight; four leaves the all-gather of a frame three frame times to complete)
    ring_streams, ring_ctxs = [], []
    if multi:
        for _ in range(4):
            ts = torch.cuda.Stream()
            ring_streams.append(ts)
            c = capi.Context(n, w, h, device_id=local_rank, stream=ts.cuda_stream,
                             flags=flags | (capi.FLAG_BLOCK_CULL if FINALIZE[0] else 0))
            upload_scene(c, n, seed, deg)
            ring_ctxs.append(c)
        ctx = ring_ctxs[0]
    else:
        ctx = capi.Context(n, w, h, device_id=-1, flags=flags)
        upload_scene(ctx, n, seed, deg)

    sr = None
    if multi:
        from godotgaussiansplatting_amd.distributed import StripeRasterizer
        # Morton-ordered scene + block culling: a rank skips the 512-splat blocks that cannot reach its stripe and
        # learns the frame's highest populated tile (quirk Q5/Q6) through a 4-byte all-reduce(MAX) per frame
        sr = StripeRasterizer(ring_ctxs, w, h, rank, world, axis=args.axis, sync_after_render=False,
                              streams=ring_streams, exchange_last_tile=FINALIZE[0])

        def step():
            sr.render_pipelined(frame)  # gather of frame k overlaps the compute of frame k+1

        def sync():
            sr.flush_all()              # the last frames are assembled inside the timed region
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        extra = []
        for _ in range(max(1, args.frames_in_flight) - 1):
            c2 = capi.Context(n, w, h, flags=flags)
            upload_scene(c2, n, seed, deg)
            extra.append(c2)
        ring = [ctx] + extra
        turn = [0]

        def step():
            ring[turn[0] % len(ring)].render(frame)
            turn[0] += 1

        def sync():
            for c in ring:
                c.synchronize()

        if len(ring) > 1:  # the strictly sequential rate, measured first on the first context alone
            for _ in range(args.warmup):
                ctx.render(frame)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx.render(frame)
            ctx.synchronize()
            sequential_fps = args.steps / (time.perf_counter() - t0)

    for i in range(args.warmup):
        step()
        if sr is not None and not args.no_rebalance and i == min(2, args.warmup - 1):
            sr.flush_all()
            sr.rebalance()  # equalise stripe cost from the measured per-column pair counts
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    fps = args.steps / elapsed
    result = {
        "metric": "frames/sec + ms/pass (proj/sort/raster) at 1080p, N-splat scene, 1/2/4/8 GPUs",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: synthetic {n:,} splats SH deg {deg} (SURVEY.md §8d generator, seed {seed}), "
                               f"{w}x{h}, fixed camera, frame left in HBM",
                   "splats": n, "width": w, "height": h, "sh_degree": deg,
                   "parallelism": "single GPU" if not multi else f"tile-{args.axis} stripes x{world} + RCCL all-gather",
                   "exp": "hardware v_exp_f32" if args.fast_exp else "contract polynomial (bit-exact vs oracle)",
                   "frames_in_flight": max(1, args.frames_in_flight) if not multi else len(ring_ctxs),
                   "scene_layout": "morton (gsplat_finalize_scene)" if FINALIZE[0] else "file order"},
    }

    # ---- per-pass and per-kernel timing (separate frames, HIP events on the context's stream) -----------------
    if rank == 0 or multi:
        ctx.set_timing(capi.FLAG_TIMING | capi.FLAG_KERNEL_TIMING)
        reps = 20
        passes, kernels, launches = [], [], None
        st = None
        for _ in range(reps):
            if sr is not None:
                sr._turn = 0  # keep the timing frames on ctx (the context whose events are read)
                sr.render(frame, assemble=False)
            else:
                ctx.render(frame)
            st = ctx.stats()
            passes.append([st["ms_projection"], st["ms_sort"], st["ms_boundaries"], st["ms_render"], st["ms_total"]])
            kernels.append([st["ms_kernel"][k] for k in st["ms_kernel"]])
            launches = st["launches_kernel"]
        ctx.set_timing(0)
        pm = np.median(np.array(passes), axis=0)
        km = dict(zip(st["ms_kernel"].keys(), np.median(np.array(kernels), axis=0)))
        st["_tiles"] = ((w + 15) // 16) * ((h + 15) // 16)
        st["_pixels"] = w * h
        if rank == 0:
            pb = phase_algorithmic_bytes(st)
            result["ms_per_pass"] = {"projection": float(pm[0]), "sort": float(pm[1]), "boundaries": float(pm[2]),
                                     "render": float(pm[3]), "gpu_total": float(pm[4])}
            result["hbm_roofline_per_pass"] = {
                k: {"algorithmic_GB": pb[k] / 1e9, "achieved_GBps": pb[k] / 1e6 / max(ms, 1e-6),
                    "frac": pb[k] / 1e6 / max(ms, 1e-6) / HBM_PEAK_GBPS}
                for k, ms in zip(["projection", "sort", "boundaries", "render"], pm[:4])}
            sr_ms = float(pm[1] + pm[3])
            result["hbm_roofline_sort_plus_raster_frac"] = (pb["sort"] + pb["render"]) / 1e6 / max(sr_ms, 1e-6) / HBM_PEAK_GBPS
            result["scene_stats"] = {"N": st["num_splats"], "V": st["num_visible"], "D": st["num_sorted"],
                                     "D_c": st["num_composited"], "overflow": st["overflow"],
                                     "sort_passes": st["sort_passes"], "sh_degree": st["sh_degree"],
                                     "sh_colours_by": "compositor (staged pairs)" if st.get("lazy_colors") else "projection pass (visible splats)",
                                     "device_bytes": st["bytes_allocated"]}
            result["ms_per_kernel_class"] = {k: float(v) for k, v in km.items()}
            if not multi:
                kb = kernel_algorithmic_bytes(st)
                # dominant kernel = longest per frame; kernels within 10 % of the longest count as tied and the one moving
                # the most algorithmic bytes is reported (on c3 projection and compositing are within 3 % of each other
                # and swap places from box to box; the compositor is VALU-bound, its HBM fraction says nothing)
                longest = max(km[k] for k in km if k in kb)
                dom = max((k for k in km if k in kb and km[k] >= 0.90 * longest), key=lambda k: kb[k])
                result["roofline_per_kernel_class"] = {
                    k: {"ms_per_frame": float(km[k]), "launches": launches[k],
                        "algorithmic_GB_per_launch": kb[k] / 1e9,
                        "achieved_GBps": kb[k] / 1e6 / max(km[k] / max(launches[k], 1), 1e-9),
                        "frac_of_hbm_peak": kb[k] / 1e6 / max(km[k] / max(launches[k], 1), 1e-9) / HBM_PEAK_GBPS}
                    for k in km if k in kb and launches[k]}
                per_launch_ms = km[dom] / max(launches[dom], 1)
                achieved = kb[dom] / 1e6 / max(per_launch_ms, 1e-9)  # GB/s
                traffic = None
                pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(pmc_path):
                    try:
                        pmc = json.load(open(pmc_path))
                        ent = pmc.get(args.config, {}).get(dom)
                        traffic = ent.get("hbm_bytes_per_launch") if ent else None
                    except Exception:
                        traffic = None
                result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                                      "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                                      "algorithmic_bytes_per_launch": kb[dom], "launches_per_frame": launches[dom],
                                      "avg_launch_ms": per_launch_ms,
                                      "measured": "HIP events on the context's stream, median of 20 frames rendered one "
                                                  "at a time after the timed region (kernel alone on the chip); with 2 "
                                                  "frames in flight the same launches overlap another frame's kernels"}

    if sequential_fps is not None:
        result["sequential_fps"] = sequential_fps  # one frame at a time on one context (frames_in_flight = 1)
    if rank == 0 and not multi and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.config, vp, cam_pos)

    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if not multi:
        ctx.close()
        for c in extra:
            c.close()
    else:
        for c in ring_ctxs:
            c.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
