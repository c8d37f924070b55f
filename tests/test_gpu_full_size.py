"""BASELINE.json configurations at their full workload sizes, through the C ABI.

c3 (the headline: 6.13 M splats, SH deg 3, 1080p), c3d (the same scene at the density of a real capture, D/N ~ 9.4)
and c4 (5.83 M splats, 4K) are compared stage by stage with the CPU oracle — the oracle needs about a second per frame
on the GPU box's host; c5 (30 M splats, 4K) is checked through size-independent properties of every stage.
Match: gsplat_projection.glsl:150-227, radix_sort_downsweep.glsl:178-213 (stable LSD contract),
gsplat_boundaries.glsl:23-50, gsplat_render.glsl:50-111.
"""
import gc

import numpy as np
import pytest

from conftest import godot_perspective

pytestmark = pytest.mark.gpu

RGBA_TOL = 1e-4


def _config_case(name):
    import oracle
    from godotgaussiansplatting_amd import scenes
    n, deg, w, h, seed = scenes.CONFIGS[name]
    rows = scenes.config_rows(name)
    records = oracle.records_from_ply_rows(rows, -10.0)
    cam = scenes.default_camera()
    proj = godot_perspective(cam.fov, w / h, cam.near, cam.far)
    vp = oracle.pack_camera(cam.xform12(), proj)
    cam_pos = np.array([-cam.origin[0], -cam.origin[1], cam.origin[2]], np.float32)
    return {"n": n, "w": w, "h": h, "rows": rows, "records": records, "vp": vp, "cam_pos": cam_pos}


def _assert_full_frame_parity(ctx, img, ref):
    st = ctx.stats()
    assert st["overflow"] == 0 and ref["stats"]["overflow"] == 0
    assert st["num_visible"] == ref["stats"]["visible"]
    assert st["num_emitted"] == ref["stats"]["emitted"]
    assert st["num_sorted"] == ref["D"]
    np.testing.assert_array_equal(ctx.read_counts(), ref["counts"])
    vis = ref["counts"] > 0
    np.testing.assert_array_equal(ctx.read_culled()[vis], ref["culled"][vis])
    sk, sv = ctx.read_sorted()
    np.testing.assert_array_equal(sk, ref["keys"])
    np.testing.assert_array_equal(sv, ref["values"])
    np.testing.assert_array_equal(ctx.read_bounds(), ref["bounds"])
    assert float(np.max(np.abs(img - ref["image"]))) <= RGBA_TOL
    np.testing.assert_array_equal(img, ref["image"])
    assert st["num_composited"] == ref["stats"]["composited"]


@pytest.mark.parametrize("name", ["c3", "c3m", "c3d", "c3r"])
def test_config3_full_size_1080p(name):
    """BASELINE.json configs[2] at workload size (the configuration the >= 1000 fps target is quoted on), the same
    scene with 4x (c3m: D/N ~ 4, part of the tiles saturating) and 7.8x (c3d: D/N ~ 9.4, a real capture's density)
    larger splats, and c3r — the same N, frame and camera with the statistics of a trained capture (surfaces, flat
    anisotropic splats, log-normal sizes with a heavy tail, bimodal opacity, floaters; D/N ~ 8, up to ~10^5 pairs in a
    tile, thousands of rectangles over 512 tiles): every stage array_equal to the oracle."""
    import oracle
    from godotgaussiansplatting_amd import capi
    c = _config_case(name)
    fr = oracle.Frame.make(c["vp"], c["cam_pos"], c["w"], c["h"])
    ref = oracle.render_frame(c["records"], fr)
    with capi.Context(c["n"], c["w"], c["h"]) as ctx:
        step = 1 << 20
        for first in range(0, c["n"], step):  # device-side ingest of the raw rows, like bench.py
            ctx.upload_ply_rows(c["rows"][first:first + step], first=first, load_time=-10.0)
        np.testing.assert_allclose(ctx.read_records()[::997], c["records"][::997], rtol=1e-6, atol=0)
        ctx.upload_splats(c["records"])  # bit-identical records for the stage-by-stage comparison
        frame = capi.make_frame(c["vp"], c["cam_pos"])
        for _ in range(3):  # the colour mode settles on the frame history
            img = ctx.render_to_host(frame)
        _assert_full_frame_parity(ctx, img, ref)
        if name == "c3":
            _twin_whole_frame(ctx, c, img, "c3")
    del ref, c
    gc.collect()


def _twin_whole_frame(ctx, c, img, name, radix_stripe=None):
    """The same frame against the INDEPENDENT float64 literal-GLSL twin (tests/test_gpu_twin.py, oracle/numpy_twin.py) —
    over the WHOLE workload-size frame, not a crop:
      * every splat's survival, tile rectangle and depth code (gsplat_projection.glsl:150-226);
      * the frame's pairs, in the reference's emission order, through the literal invocation-by-invocation emulation of
        radix_sort_{upsweep,spine,downsweep}.glsl at subgroup size 32 (oracle/radix_glsl.py) — all of them, or
        (radix_stripe = (x0, x1) tile columns) one stripe's worth: the restriction of a stable sort to a subset of the keys
        is the stable sort of the subset;
      * RasterizeData of every splat any tile lists, to binary32 rounding (gsplat_projection.glsl:202-206);
      * every pixel within the north-star 1e-4 of the twin's compositor on the frame's own (binary32) records and tile
        lists, except knife-edge pixels — found by the twin alone — and the end-to-end error on the twin's own float64
        records, reported with a loose bound (gsplat_render.glsl:50-101).
    The report goes to gpurun_out/twin_report_<name>.json (committed under profiles/)."""
    import json
    import os
    import time
    import twin_checks as tc
    from oracle import radix_glsl as rg
    t0 = time.time()
    w, h = c["w"], c["h"]
    gx, gy = (w + 15) // 16, (h + 15) // 16
    counts = ctx.read_counts()
    sk, sv = ctx.read_sorted()
    bounds = ctx.read_bounds()
    p = tc.project_chunked(c["records"], c["vp"], c["cam_pos"], 1.0, w, h)
    rep = tc.check_integer_decisions(p, counts, sk, sv, c["n"])
    assert rep["compared_rects"] > 0.99 * rep["visible"] and rep["unstable_cull_or_rect_frac"] < 5e-3
    own_p = {k: p[k] for k in ("alive", "rect", "depth16", "gx", "gy")}   # (kept for the own-order sample below)
    del p
    # the literal sort shaders on the frame's pairs in emission order (ascending splat id, y outer / x inner)
    if radix_stripe is None:
        sel = np.ones(sk.size, bool)
    else:
        col = (sk >> 16) % gx
        sel = (col >= radix_stripe[0]) & (col < radix_stripe[1])
    ssk, ssv = sk[sel], sv[sel]
    order = np.lexsort((np.arange(ssk.size), ssv))
    lk, lv = rg.sort_pairs(ssk[order], ssv[order])
    np.testing.assert_array_equal(lk, ssk)
    np.testing.assert_array_equal(lv, ssv)
    rep["pairs_through_the_literal_sort_shaders"] = int(ssk.size)
    del lk, lv, order, ssk, ssv, sel
    full = (0, gx, 0, gy)
    ids = tc.splats_in_tiles(sv, bounds, gx, full)
    culled = ctx.read_culled()
    want = tc.twin_records_of(c["records"], c["vp"], c["cam_pos"], 1.0, w, h, 0.0, ids)
    worst = tc.records_error(culled[ids], want, w, h)
    own = np.zeros((c["n"], 12))
    own[ids] = want
    del want
    rep["records"] = {"records_compared": int(ids.size), "max_rel_err": worst}
    rep["image"] = tc.check_image_full(culled, w, h, 0.0, img, sv, bounds)
    rep["image_end_to_end"] = tc.check_image_full(own, w, h, 0.0, img, sv, bounds, tol=5e-3)
    # ... and where that end-to-end error comes from: the same float64 records with ONE thing taken from the frame — the
    # splat centres (image_pos, two binary32 numbers per record; one ulp at x > 1024 is 1.2e-4 px and a sub-pixel splat
    # turns that into ~1e-3 of alpha).  With the centres the reference's compositor would read (it reads a binary32 buffer
    # too) everything else in float64 — covariance inversion, colours, opacity, and the producer's depth ORDER — the frame
    # is within the north star's 1e-4 off knife edges: the 3e-3 above is the rounding of image_pos, nothing else
    own[ids, 0:2] = np.asarray(culled[ids, 0:2], np.float64)
    rep["image_end_to_end_binary32_centres"] = tc.check_image_full(own, w, h, 0.0, img, sv, bounds, tol=RGBA_TOL)
    # ... and the one place where the witness still borrowed the answer: the ORDER.  >= 1 % of the populated tiles, composited
    # from the twin's OWN member lists (its float64 rectangles) in its OWN depth order (its float64 depth codes, ties by splat
    # id) on those same records (float64, binary32 centres): how many pixels differ from the GPU frame by more than 1e-4,
    # by tile class (identical list / same set in another order = depth codes that floor differently / another set)
    rep["own_order_sample"] = tc.own_order_sample(own_p, own, img, sv, bounds, w, h)
    oo = rep["own_order_sample"]["per_class"]
    assert rep["own_order_sample"]["tiles_sampled"] >= 0.01 * rep["own_order_sample"]["of_populated_tiles"]
    # where the twin's list IS the producer's, nothing but knife edges may exceed the tolerance
    assert oo["identical_list"]["pixels_over_tol"] == oo["identical_list"]["of_them_on_knife_edges"], oo
    rep["frame"] = {"config": name, "width": w, "height": h, "splats": c["n"], "pairs": int(sk.size), "tiles": gx * gy}
    rep["host_seconds"] = round(time.time() - t0, 1)
    print(name, "twin", json.dumps(rep))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"twin_report_{name}.json"), "w") as f:
            json.dump(rep, f, indent=1)


def test_config4_full_size_4k():
    """BASELINE.json configs[3] at workload size: one full-frame context vs the oracle, then the same frame as 8 tile
    column stripes rendered one after the other on this GPU — their union must be the same image and tile ranges."""
    import oracle
    from godotgaussiansplatting_amd import capi
    c = _config_case("c4")
    fr = oracle.Frame.make(c["vp"], c["cam_pos"], c["w"], c["h"])
    ref = oracle.render_frame(c["records"], fr)
    gx, gy = oracle.grid(c["w"], c["h"])
    frame = capi.make_frame(c["vp"], c["cam_pos"])
    with capi.Context(c["n"], c["w"], c["h"]) as ctx:
        ctx.upload_splats(c["records"])
        for _ in range(3):
            img = ctx.render_to_host(frame)
        _assert_full_frame_parity(ctx, img, ref)
        # the compositor's XCD-local heaviest-first schedule covers a 32 400-tile frame too (eight lists of 4 080 slots):
        # the whole table against tests/schedule_model.py, built from the staged counts of the frame before
        import schedule_model as sm
        prev = ctx.read_tile_staged()
        ctx.render(frame)
        entries = sm.order_layout(gx, gy)["entries"]
        order = ctx.read_tile_order(entries)
        assert order.size == entries
        np.testing.assert_array_equal(np.sort(order[order != sm.EMPTY]), np.arange(gx * gy, dtype=np.uint32))
        np.testing.assert_array_equal(order, sm.expected_order(prev, (0, gx, 0, gy), gx, "xcd"))
        ctx.render(frame)  # (the taps of the twin check below belong to a full frame again)
        _twin_whole_frame(ctx, c, img, "c4")   # (all 13.4 M pairs through the literal sort shaders; round 4: one sparse stripe)
        # 8 column stripes (balanced by tile count): every stripe's tiles, pixels and pair count
        edges = [round(gx * k / 8) for k in range(9)]
        union = np.zeros_like(img)
        bounds_ok = np.zeros(gx * gy, bool)
        total_pairs = 0
        ref_tiles = ref["keys"] >> 16
        for k in range(8):
            x0, x1 = edges[k], edges[k + 1]
            ctx.set_stripe(capi.STRIPE_COLUMNS, x0, x1)
            simg = ctx.render_to_host(frame)
            st = ctx.stats()
            total_pairs += st["num_sorted"]
            px0, px1 = x0 * 16, min(x1 * 16, c["w"])
            union[:, px0:px1] = simg[:, px0:px1]
            sk, sv = ctx.read_sorted()
            cols = ref_tiles % gx
            sel = (cols >= x0) & (cols < x1)
            np.testing.assert_array_equal(sk, ref["keys"][sel])    # the stripe's pairs = the frame's pairs of its tiles,
            np.testing.assert_array_equal(sv, ref["values"][sel])  # in the same order
            b = ctx.read_bounds().astype(np.int64)
            rb = ref["bounds"].astype(np.int64)
            for ty in range(0, gy):
                t = np.arange(ty * gx + x0, ty * gx + x1)
                # ranges are stripe-local offsets: lengths must agree tile by tile (Q5/Q6 included)
                np.testing.assert_array_equal(np.maximum(b[t, 1] - b[t, 0], 0), np.maximum(rb[t, 1] - rb[t, 0], 0))
                bounds_ok[t] = True
        assert bounds_ok.all() and total_pairs == ref["D"]
        np.testing.assert_array_equal(union, ref["image"])
    del ref, c
    gc.collect()


def _bench_like_row_cuts(views, frame, gy, world, top_ptr, prior):
    """The stripe boundaries bench.py --gpus N arrives at (rebalance_groups), on one GPU: cuts that equalise the pairs per
    tile row (+ a constant per tile), then three rounds of distributed.time_balanced_cuts from every rank's MEASURED frame
    time under the current cuts (five frames each, median of gsplat_stats.ms_total)."""
    from godotgaussiansplatting_amd import capi
    from godotgaussiansplatting_amd.distributed import balanced_cuts, time_balanced_cuts
    cuts = balanced_cuts(prior, world)
    log = []
    for _ in range(3):
        times = []
        for r, v in enumerate(views):
            v.set_stripe(capi.STRIPE_ROWS, cuts[r], cuts[r + 1])
            v.set_timing(capi.FLAG_TIMING)
            ms = []
            for _k in range(5):
                v.render_begin(frame)
                v.render_end(frame_last_tile_ptr=top_ptr)
                ms.append(v.stats()["ms_total"])
            v.set_timing(0)
            times.append(float(np.median(ms)))
        log.append([round(t, 4) for t in times])
        cuts = time_balanced_cuts(cuts, times, prior=prior)
    return [int(x) for x in cuts], log


@pytest.mark.parametrize("name", ["c3", "c4"])
def test_multi_gpu_default_configuration_at_workload_size(name):
    """The configuration `bench.py --gpus N` renders with — Morton layout (gsplat_finalize_scene) + GSPLAT_FLAG_BLOCK_CULL +
    GSPLAT_FLAG_TIES_STORAGE_ORDER + the one-pass pair sort on stripe-local 16-bit tile ids + 8 ROW stripes cut like the
    bench cuts them (pairs-balanced, then re-cut three times from measured rank times), every stripe rendered through
    gsplat_render_begin / gsplat_render_end with the FRAME's last tile handed over (the group's protocol: MAX over the
    ranks' words) — against the ORACLE at workload size.  The storage-order contract's frame is the default contract's
    frame of the same scene uploaded in storage order (GSPLAT_DEBUG_SLOT_IDS), so the oracle renders records[id_of_slot]
    once; every stripe's sorted keys, values (through the slot -> id map), tile-range lengths and the frame assembled from
    the eight stripes must be array_equal — the same tie order the first 8-GPU headline will be rendered under.
    Match: gsplat_projection.glsl:196,218-226, gsplat_boundaries.glsl:39-49, gsplat_render.glsl:50-111; SURVEY.md §8(e)."""
    import torch
    import oracle
    from godotgaussiansplatting_amd import capi
    world = 8
    c = _config_case(name)
    n, w, h = c["n"], c["w"], c["h"]
    gx, gy = oracle.grid(w, h)
    flags = capi.FLAG_BLOCK_CULL | capi.FLAG_TIES_STORAGE_ORDER
    frame = capi.make_frame(c["vp"], c["cam_pos"])
    owner = capi.Context(n, w, h, flags=flags)
    views = [owner]
    try:
        owner.upload_splats(c["records"])
        owner.finalize_scene()
        id_of_slot = owner.read_slot_ids()
        assert not np.array_equal(id_of_slot, np.arange(n, dtype=np.uint32))
        fr = oracle.Frame.make(c["vp"], c["cam_pos"], w, h)
        ref = oracle.render_frame(c["records"][id_of_slot], fr)          # the scene in storage order, default contract
        ref_values = id_of_slot[ref["values"]]                            # ... its values as splat ids (what the tap speaks)
        assert ref["stats"]["overflow"] == 0
        ties = ref["keys"][1:] == ref["keys"][:-1]
        default_values_differ = bool(np.any(ref_values[1:][ties] < ref_values[:-1][ties]))
        assert ties.sum() > 1000 and default_values_differ, "the frame must hold ties the two contracts resolve differently"
        # the full-frame context with these flags (what bench.py's frame_equal compares the assembled frame with)
        for _ in range(3):
            img = owner.render_to_host(frame)
        st = owner.stats()
        assert st["pair_key_bytes"] == 2 and st["num_sorted"] == ref["D"] and st["overflow"] == 0
        sk, sv = owner.read_sorted()
        np.testing.assert_array_equal(sk, ref["keys"])
        np.testing.assert_array_equal(sv, ref_values)
        np.testing.assert_array_equal(owner.read_bounds(), ref["bounds"])
        np.testing.assert_array_equal(img, ref["image"])
        del sk, sv, img
        # eight members = eight views on the one scene, row stripes
        views += [owner.view(flags=flags) for _ in range(world - 1)]
        rb = ref["bounds"].astype(np.int64)
        cnt = np.clip(rb[:, 1] - rb[:, 0], 0, None).reshape(gy, gx)
        prior = cnt.sum(axis=1).astype(np.float64) + 64.0 * gx
        top = torch.zeros(1, dtype=torch.int32, device="cuda")
        top.fill_(int(ref["keys"][-1] >> 16) + 1)
        torch.cuda.synchronize()
        cuts, log = _bench_like_row_cuts(views, frame, gy, world, top.data_ptr(), prior)
        assert cuts[0] == 0 and cuts[-1] == gy and all(b > a for a, b in zip(cuts[:-1], cuts[1:]))
        print(name, "row cuts", cuts, "rank ms per re-cut round", log)
        # one frame the way gsplat_group_render runs it: begin on every member, MAX of the members' words, end on every member
        words = torch.zeros(world, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for r, v in enumerate(views):
            v.set_stripe(capi.STRIPE_ROWS, cuts[r], cuts[r + 1])
        for rep in range(2):        # (the second frame runs on the hints of the first: emit / sort forms settled)
            for r, v in enumerate(views):
                v.render_begin(frame, words[r:r + 1].data_ptr())
            for v in views:
                v.synchronize()
            local = words.cpu().numpy().copy()
            top = words.max().reshape(1).contiguous()
            torch.cuda.synchronize()
            for v in views:
                v.render_end(frame_last_tile_ptr=top.data_ptr())
            for v in views:
                v.synchronize()
        assert int(top.item()) == int(ref["keys"][-1] >> 16) + 1            # the frame's highest populated tile + 1
        # (whether a stripe sees the frame's last tile by itself depends on the scene: c3's splats next to the camera fill the
        # screen and reach tile T - 1 from every stripe, c4's edge stripes do not — the exchange is exercised either way, and
        # test_block_cull_stripes_with_last_tile_exchange asserts the case where it matters)
        print(name, "members' own last tile + 1:", local.tolist(), "frame's:", int(top.item()))
        out = np.full_like(ref["image"], -1.0)
        tile_row = (ref["keys"] >> 16) // gx
        total, one_pass, skipped = 0, 0, 0
        for r, v in enumerate(views):
            y0, y1 = cuts[r], cuts[r + 1]
            st = v.stats()
            sel = (tile_row >= y0) & (tile_row < y1)
            assert st["num_sorted"] == int(sel.sum()) and st["pair_key_bytes"] == 2 and st["overflow"] == 0
            total += st["num_sorted"]
            stripe_tiles = (y1 - y0) * gx
            # the one-pass pair sort ran wherever the stripe allows it (<= 4096 stripe-local tile ids): 2 splat passes + 1
            if 256 < stripe_tiles <= 4096:
                assert st["sort_passes"] == 3, (r, stripe_tiles, st["sort_passes"])
                one_pass += 1
            sk, sv = v.read_sorted()
            np.testing.assert_array_equal(sk, ref["keys"][sel], err_msg=f"keys of rank {r}")
            np.testing.assert_array_equal(sv, ref_values[sel], err_msg=f"values of rank {r}")
            b = v.read_bounds().astype(np.int64)
            t = np.arange(y0 * gx, y1 * gx)
            # ranges are stripe-local offsets: lengths must agree tile by tile (Q5/Q6 included)
            np.testing.assert_array_equal(np.maximum(b[t, 1] - b[t, 0], 0), np.maximum(rb[t, 1] - rb[t, 0], 0),
                                          err_msg=f"tile ranges of rank {r}")
            py0, py1 = y0 * 16, min(y1 * 16, h)
            out[py0:py1] = v.read_image()[py0:py1]
            skipped += int(v.read_block_sums()[:, 3].sum())
            del sk, sv, b
        assert total == ref["D"]
        assert one_pass >= world // 2, f"only {one_pass} of {world} stripes took the one-pass pair sort (cuts {cuts})"
        assert skipped > 0.3 * world * ((n + 511) // 512), "the stripes must really skip blocks"
        np.testing.assert_array_equal(out, ref["image"])
    finally:
        for v in reversed(views):
            v.close()
    del ref, c
    gc.collect()


def test_config5_properties():
    """BASELINE.json configs[4] at workload size (30 M splats, 4K, the radix-sort stress) on one GPU: sortedness,
    permutation of the emitted pairs, ties in ascending splat id, tile ranges partition the sorted array, alpha == 1,
    no overflow, and the per-splat tile counts add up to D."""
    from godotgaussiansplatting_amd import capi, scenes
    n, deg, w, h, seed = scenes.CONFIGS["c5"]
    cam = scenes.default_camera()
    vp, cam_pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far)
    frame = capi.make_frame(vp, cam_pos)
    rows = scenes.config_rows("c5")
    with capi.Context(n, w, h, flags=capi.FLAG_KEEP_EMITTED) as ctx:
        step = 1 << 21
        for first in range(0, n, step):
            ctx.upload_ply_rows(rows[first:first + step], first=first, load_time=-10.0)
        del rows
        gc.collect()
        ctx.render(frame)
        ctx.render(frame)
        ctx.synchronize()
        st = ctx.stats()
        d = st["num_sorted"]
        assert st["overflow"] == 0 and st["num_emitted"] == d and d > 40_000_000
        counts = ctx.read_counts()
        assert int(counts.sum(dtype=np.uint64)) == d and int((counts > 0).sum()) == st["num_visible"]
        sk, sv = ctx.read_sorted()
        assert sk.size == d and np.all(sk[1:] >= sk[:-1])                                   # sorted
        ek, ev = ctx.read_emitted()
        pe = np.sort((ek.astype(np.uint64) << np.uint64(32)) | ev)
        ps = np.sort((sk.astype(np.uint64) << np.uint64(32)) | sv)
        assert np.array_equal(pe, ps)                                                       # permutation of the emission
        del pe, ps, ek, ev
        same = sk[1:] == sk[:-1]
        assert np.all(sv[1:][same] > sv[:-1][same])                                         # stable: ties by splat id
        assert np.array_equal(np.bincount(sv, minlength=n).astype(np.uint32), counts)       # every splat's pairs survive
        gx, gy = (w + 15) // 16, (h + 15) // 16
        tiles = sk >> 16
        assert int(tiles.max()) < gx * gy
        b = ctx.read_bounds().astype(np.int64)
        first = np.flatnonzero(np.r_[True, tiles[1:] != tiles[:-1]])
        t_ids = tiles[first]
        ends = np.r_[first[1:], d]
        last_t = int(t_ids[-1])
        inner = t_ids != last_t
        assert np.array_equal(b[t_ids[inner], 0], first[inner]) and np.array_equal(b[t_ids[inner], 1], ends[inner])
        # quirk Q5/Q6 of gsplat_boundaries.glsl:39-49 on the highest populated tile
        assert b[last_t, 0] == first[-1] and b[last_t, 1] == ((d - 1) if last_t == gx * gy - 1 else 0)
        empty = np.ones(gx * gy, bool)
        empty[t_ids] = False
        assert not b[empty].any()
        img = ctx.read_image()
        assert np.all(img[..., 3] == 1.0) and np.isfinite(img).all()                        # gsplat_render.glsl:101
        assert img[..., :3].min() >= 0.0 and img[..., :3].max() > 0.1
        staged = ctx.read_tile_staged().astype(np.int64)
        n_t = np.maximum(b[:, 1] - b[:, 0], 0)
        assert np.all(staged <= n_t) and np.all((staged == n_t) | (staged % 256 == 0))      # batches of 256 until the exit
