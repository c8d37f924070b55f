"""A producer of frames (the HIP path on the GPU; the C oracle in the CPU tests) against the float64 literal-GLSL twin
(oracle/numpy_twin.py), which was written from the shader text alone and shares no code and no arithmetic shortcuts
with either.  Staged at the reference's own buffer boundaries:
  * integer decisions (survival, tile rectangles, depth codes): bit-exact wherever the twin's own perturbation analysis
    says a binary32 evaluation has no choice;
  * RasterizeData (a binary32 buffer in the reference as well, gsplat_projection.glsl:42-48): to binary32 rounding;
  * the image: the twin's compositor on the producer's RasterizeData and tile lists, within the north-star 1e-4 off
    the knife-edge pixels — and which pixels those are is again decided by the twin alone.
(An end-to-end bound of 1e-4 against ideal arithmetic is not a property ANY binary32 evaluation has: one ulp of
image_pos at x > 1024 is 1.2e-4 px, and a sub-pixel splat turns that into ~1e-3 of alpha.  The end-to-end error is
reported, with a loose bound.)  Returns small reports (fractions of unstable decisions / knife-edge pixels, errors)."""
import numpy as np

from oracle import numpy_twin as twin

RGBA_TOL = 1e-4   # BASELINE.json north_star: per channel


_PROJ = {}


def _project_part(span):
    g = _PROJ
    s, e = span
    rec = g["records"][s:e] if g["ids"] is None else g["records"][g["ids"][s:e]]
    p = twin.project(rec, g["vp"][:16], g["vp"][16:], g["cam_pos"], g["model_scale"], g["w"], g["h"], time=g["time"])
    if g["want"] == "raster":
        return s, e, p["raster"]
    return s, e, {k: p[k] for k in ("alive", "stable_rect", "stable_alive", "stable_depth", "rect", "count", "depth16", "gx", "gy")}


def _project_parallel(records, vp, cam_pos, model_scale, w, h, time, chunk, ids, want, procs):
    """twin.project over the splats (all, or `ids`) a chunk at a time, the chunks spread over forked workers (NumPy's
    float64 evaluation of ~150 expressions per splat is single-threaded: 2 us per splat in cache-sized chunks)."""
    import multiprocessing as mp
    import os
    n = records.shape[0] if ids is None else ids.size
    spans = [(s, min(s + chunk, n)) for s in range(0, n, chunk)]
    _PROJ.update(records=records, vp=vp, cam_pos=cam_pos, model_scale=model_scale, w=w, h=h, time=time, ids=ids, want=want)
    # (forking and pickling the results back costs about as much as 30 chunks of work)
    procs = procs if procs is not None else (max(1, min(16, (os.cpu_count() or 2) - 1)) if len(spans) >= 32 else 1)
    try:
        if procs <= 1 or len(spans) <= 1:
            for sp in spans:
                yield _project_part(sp)
        else:
            with mp.get_context("fork").Pool(procs) as pool:
                for r in pool.imap_unordered(_project_part, spans):
                    yield r
    finally:
        _PROJ.clear()


def project_chunked(records, vp, cam_pos, model_scale, w, h, time=0.0, chunk=50_000, procs=None):
    """twin.project over all splats, a chunk at a time; per-splat decisions only (no float records kept)."""
    n = records.shape[0]
    out = {k: np.zeros(n, bool) for k in ("alive", "stable_rect", "stable_alive", "stable_depth")}
    out["rect"] = np.zeros((n, 4), np.int32)
    out["count"] = np.zeros(n, np.int64)
    out["depth16"] = np.zeros(n, np.int32)
    grid = ((w + 15) // 16, (h + 15) // 16)
    for s, e, p in _project_parallel(records, vp, cam_pos, model_scale, w, h, time, chunk, None, "decisions", procs):
        for k in out:
            out[k][s:e] = p[k]
        grid = (p["gx"], p["gy"])
    out["gx"], out["gy"] = grid
    return out


def twin_records_of(records, vp, cam_pos, model_scale, w, h, time, ids, chunk=50_000, procs=None):
    """(ids.size, 12) float64: the twin's own RasterizeData of the splats `ids`, chunks in parallel."""
    want = np.zeros((ids.size, 12))
    for s, e, r in _project_parallel(records, vp, cam_pos, model_scale, w, h, time, chunk, ids, "raster", procs):
        want[s:e] = r
    return want


def rects_from_pairs(sorted_keys, sorted_values, n, gx):
    """Per splat: the bounding rectangle of its tiles, their number and its depth code, from a frame's sorted pairs."""
    tiles = (sorted_keys >> 16).astype(np.int64)
    tx, ty = tiles % gx, tiles // gx
    v = sorted_values.astype(np.int64)
    x0 = np.full(n, 1 << 30); y0 = np.full(n, 1 << 30); x1 = np.zeros(n, np.int64); y1 = np.zeros(n, np.int64)
    np.minimum.at(x0, v, tx); np.minimum.at(y0, v, ty)
    np.maximum.at(x1, v, tx + 1); np.maximum.at(y1, v, ty + 1)
    cnt = np.bincount(v, minlength=n)
    depth = np.zeros(n, np.int64)
    depth[v] = sorted_keys & 0xFFFF
    dmin = np.full(n, 1 << 30)
    np.minimum.at(dmin, v, (sorted_keys & 0xFFFF).astype(np.int64))
    assert np.array_equal(depth[cnt > 0], dmin[cnt > 0]), "a splat's pairs carry different depth codes"
    rect = np.stack([x0, y0, x1, y1], 1)
    rect[cnt == 0] = 0
    return rect, cnt, depth


def check_integer_decisions(p, counts, sorted_keys, sorted_values, n):
    """Survivors, tile rectangles and depth codes of the producer vs the twin's, where the twin calls them stable."""
    gx = p["gx"]
    alive_g = counts > 0
    rect_g, cnt_g, depth_g = rects_from_pairs(sorted_keys, sorted_values, n, gx)
    assert np.array_equal(cnt_g, counts.astype(np.int64)), "pairs per splat != tile counts tap"
    # a rectangle is a rectangle: count = w * h
    assert np.array_equal((rect_g[:, 2] - rect_g[:, 0]) * (rect_g[:, 3] - rect_g[:, 1]), cnt_g)
    stable = p["stable_alive"] & p["stable_rect"]
    diff_alive = alive_g != p["alive"]
    assert not (diff_alive & stable).any(), f"{int((diff_alive & stable).sum())} stable splats culled differently"
    both = alive_g & p["alive"] & stable
    bad = (rect_g[both] != p["rect"][both]).any(axis=1)
    assert not bad.any(), f"{int(bad.sum())} stable tile rectangles differ"
    dd = np.abs(depth_g[both] - p["depth16"][both])
    sd = p["stable_depth"][both]
    assert not dd[sd].any(), f"{int((dd[sd] != 0).sum())} stable depth codes differ"
    assert dd.max(initial=0) <= 1
    vis = max(int(alive_g.sum()), 1)
    return {"visible": int(alive_g.sum()), "culled_differently": int(diff_alive.sum()),
            "unstable_cull_or_rect_frac": float((~stable & (alive_g | p["alive"])).sum() / vis),
            "compared_rects": int(both.sum()), "unstable_depth_frac": float((~sd).mean()) if both.any() else 0.0,
            "depth_off_by_one": int((dd != 0).sum())}


def twin_records(records, vp, cam_pos, model_scale, w, h, time, ids):
    """The twin's own float64 RasterizeData of the splats `ids` (zero elsewhere)."""
    raster = np.zeros((records.shape[0], 12))
    if ids.size:
        raster[ids] = twin.project(records[ids], vp[:16], vp[16:], cam_pos, model_scale, w, h, time=time)["raster"]
    return raster


def splats_in_tiles(sorted_values, bounds, gx, tiles):
    x0, x1, y0, y1 = tiles
    b = bounds.astype(np.int64)
    need = [sorted_values[b[ty * gx + tx, 0]:b[ty * gx + tx, 1]] for ty in range(y0, y1) for tx in range(x0, x1)
            if b[ty * gx + tx, 1] > b[ty * gx + tx, 0]]
    return np.unique(np.concatenate(need)).astype(np.int64) if need else np.zeros(0, np.int64)


def records_error(got, want, w, h):
    """Largest relative difference between binary32 RasterizeData records `got` and the twin's float64 `want` (n x 12)."""
    got = np.asarray(got, np.float64)
    # each field against the magnitude of the quantity it is a component of: the three conic entries share the scale of
    # the largest (the off-diagonal one is a difference of products and may be tiny), a colour is a sum of ~16 terms of
    # order one, positions stand for themselves
    scale = np.maximum(np.abs(want), 1.0)
    scale[:, 4:7] = np.maximum(np.abs(want[:, 4:7]).max(axis=1, keepdims=True), 1.0)
    # image_pos = ((ndc + 1) / 2) * (dims - 1): a few binary32 roundings at the magnitude of the image size, also where
    # the result itself is near 0 (the left / top edge): 3e-5 * dims / 8 = 32 ulps of the width
    scale[:, 0], scale[:, 1] = max(w / 8.0, 1.0), max(h / 8.0, 1.0)
    rel = np.abs(got - want) / scale
    assert rel.max(initial=0) <= 3e-5, f"RasterizeData off by {rel.max():.3g} (relative), field {int(np.argmax(rel.max(axis=0)))}"
    return float(rel.max(initial=0))


def check_records(culled, records, vp, cam_pos, model_scale, w, h, time, ids):
    """gsplat_projection.glsl:202-206: the producer's binary32 RasterizeData of the splats `ids` vs the twin's float64
    evaluation, to binary32 rounding of the (longer) expressions behind them."""
    want = twin_records(records, vp, cam_pos, model_scale, w, h, time, ids)[ids]
    return {"records_compared": int(ids.size), "max_rel_err": records_error(culled[ids], want, w, h)}


def check_image(raster, w, h, heat, img, sorted_values, bounds, tiles, tol=RGBA_TOL):
    """gsplat_render.glsl:50-101: the twin's literal float64 compositor on the RasterizeData `raster` (the producer's
    own binary32 buffer — the reference's compositor reads a binary32 buffer too — or the twin's float64 records),
    walking the producer's tile lists, inside the rectangle of tiles `tiles` = (x0, x1, y0, y1).  Knife-edge pixels
    (the t <= 1/255 stop, the block early-exit sum) are found by the twin alone: its own result under a +-4e-6
    perturbation of every alpha."""
    x0, x1, y0, y1 = tiles
    raster = np.asarray(raster, np.float64)
    kw = dict(heatmap_factor=heat, tiles=tiles)
    base = twin.render(raster, sorted_values, bounds, w, h, **kw)
    lo = twin.render(raster, sorted_values, bounds, w, h, alpha_scale=1 - 4e-6, **kw)
    hi = twin.render(raster, sorted_values, bounds, w, h, alpha_scale=1 + 4e-6, **kw)
    px0, px1, py0, py1 = x0 * 16, min(x1 * 16, w), y0 * 16, min(y1 * 16, h)
    sl = (slice(py0, py1), slice(px0, px1))
    knife = np.max(np.abs(hi[sl] - lo[sl]), axis=-1) > 2e-5
    err = np.max(np.abs(img[sl].astype(np.float64) - base[sl]), axis=-1)
    assert knife.mean() < 0.03, f"knife-edge pixels {knife.mean():.4f}"
    assert err[~knife].max(initial=0) <= tol, f"max |rgba - twin| off knife edges = {err[~knife].max():.3g}"
    assert np.all(img[sl][..., 3] == 1.0)
    assert float(base[sl][..., :3].max()) > 0.05, "the crop is empty"
    return {"pixels": int(knife.size), "knife_edge_frac": float(knife.mean()),
            "max_err_off_knife_edges": float(err[~knife].max(initial=0)),
            "max_err_on_knife_edges": float(err[knife].max(initial=0))}


# ---- the whole frame: the twin's compositor is a Python loop over tiles and staged splats (~20 us per step), so a
# 1080p / 4K frame at workload size is spread over the host's cores, one tile row per task (fork: the records and the
# tile lists are shared copy-on-write) -------------------------------------------------------------------------------
_FULL = {}


def _full_row(by):
    g = _FULL
    w, h, gx = g["w"], g["h"], g["gx"]
    kw = dict(heatmap_factor=g["heat"], tiles=(0, gx, by, by + 1))
    y0, y1 = by * 16, min(by * 16 + 16, h)
    out = []
    for scale in (1.0, 1 - 4e-6, 1 + 4e-6):
        out.append(twin.render(g["raster"], g["sv"], g["bounds"], w, h, alpha_scale=scale, **kw)[y0:y1].astype(np.float64))
    return by, out


def check_image_full(raster, w, h, heat, img, sorted_values, bounds, tol=RGBA_TOL, procs=None):
    """check_image over EVERY tile of the frame (same criteria, same knife-edge rule), tile rows in parallel."""
    import multiprocessing as mp
    import os
    gx, gy = (w + 15) // 16, (h + 15) // 16
    _FULL.update(raster=np.asarray(raster, np.float64), sv=sorted_values, bounds=bounds, w=w, h=h, gx=gx, heat=heat)
    base = np.zeros((h, w, 4))
    lo = np.zeros((h, w, 4))
    hi = np.zeros((h, w, 4))
    procs = procs or max(1, min(64, (os.cpu_count() or 2) - 1))
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        for by, (b, l, u) in pool.imap_unordered(_full_row, range(gy)):
            y0, y1 = by * 16, min(by * 16 + 16, h)
            base[y0:y1], lo[y0:y1], hi[y0:y1] = b, l, u
    _FULL.clear()
    knife = np.max(np.abs(hi - lo), axis=-1) > 2e-5
    err = np.max(np.abs(img.astype(np.float64) - base), axis=-1)
    assert knife.mean() < 0.03, f"knife-edge pixels {knife.mean():.4f}"
    assert err[~knife].max(initial=0) <= tol, f"max |rgba - twin| off knife edges = {err[~knife].max():.3g}"
    assert np.all(img[..., 3] == 1.0)
    rep = {"pixels": int(knife.size), "tiles": int(gx * gy), "knife_edge_frac": float(knife.mean()),
           "knife_edge_pixels": int(knife.sum()),
           "max_err_off_knife_edges": float(err[~knife].max(initial=0)),
           "max_err_on_knife_edges": float(err[knife].max(initial=0)),
           "pixels_over_tol_on_knife_edges": int((err[knife] > tol).sum())}
    # the knife-edge pixels the producer misses by more than the tolerance: WHICH discontinuity each of them sits on
    ys, xs = np.nonzero(knife & (err > tol))
    rep["knife_edge_outliers"] = [explain_pixel(raster, sorted_values, bounds, w, h, int(x), int(y), img[y, x], base[y, x])
                                  for y, x in list(zip(ys, xs))[:16]]
    return rep


def explain_pixel(raster, sorted_values, bounds, w, h, x, y, got, want):
    """Where does pixel (x, y) stop in the twin's compositor under alphas scaled by 1 - 4e-6 / 1 + 4e-6, and where does its
    TILE leave the batch loop?  A knife-edge pixel is one where the two runs part: either the pixel's own stop
    (transmittance crossing 1/255 at one list entry or the next, gsplat_render.glsl:79) or the tile's early exit (the
    block sum of uint(t * 255) crossing 255 at a batch boundary, :66,97), after which the pixel composites one more batch
    or none.  Returns both runs' stop entry / transmittance / splat id and exit batch, and which of the two decisions
    differs."""
    raster = np.asarray(raster, np.float64)
    gx = (w + 15) // 16
    bx, by = x // 16, y // 16
    tid = by * gx + bx
    b0, b1 = int(bounds[tid, 0]), int(bounds[tid, 1])
    num = max(b1 - b0, 0)
    ly, lx = np.divmod(np.arange(256), 16)
    px, py = (bx * 16 + lx).astype(np.float64), (by * 16 + ly).astype(np.float64)
    me = (y - by * 16) * 16 + (x - bx * 16)
    runs = {}
    for name, scale in (("alpha x (1 - 4e-6)", 1 - 4e-6), ("alpha x (1 + 4e-6)", 1 + 4e-6)):
        t = np.ones(256)
        stop_entry, stop_t, exit_batch, shared = None, None, None, 0xFFFFFFFF
        i = 0
        while i * 256 < num and shared > 255:
            ids = sorted_values[b0 + 256 * i: b0 + min(256 * (i + 1), num)]
            for j, sid in enumerate(ids):
                live = t > 1.0 / 255
                if not live.any():
                    break
                if not live[me] and stop_entry is None:
                    stop_entry, stop_t = 256 * i + j, float(t[me])
                rr = raster[sid]
                dx, dy = rr[0] - px, rr[1] - py
                alpha = rr[11] * np.exp(-0.5 * (rr[4] * dx * dx + rr[6] * dy * dy) - rr[5] * dx * dy) * scale
                t = np.where(live, t * (1 - alpha), t)
            shared = int(np.sum(np.floor(t * 255).astype(np.int64)))
            i += 1
        exit_batch = i
        runs[name] = {"pixel_left_its_loop_before_entry": stop_entry, "transmittance_there": stop_t,
                      "splat_id_there": int(sorted_values[b0 + stop_entry]) if stop_entry is not None and stop_entry < num else None,
                      "tile_left_after_batches": exit_batch, "block_sum_then": shared, "final_transmittance": float(t[me])}
    a, b = runs.values()
    which = ("tile early exit (block sum crosses 255 at a batch boundary)" if a["tile_left_after_batches"] != b["tile_left_after_batches"]
             else "pixel stop (t crosses 1/255)" if a["pixel_left_its_loop_before_entry"] != b["pixel_left_its_loop_before_entry"]
             else "neither differs at 4e-6: a near miss of the mask")
    return {"pixel": [x, y], "tile": int(tid), "pairs_in_tile": num, "producer_rgb": [float(v) for v in got[:3]],
            "twin_rgb": [float(v) for v in want[:3]], "max_abs_diff": float(np.max(np.abs(np.asarray(got[:3], np.float64) - want[:3]))),
            "decision": which, "runs": runs}


def own_order_sample(p, own_records, img, sorted_values, bounds, w, h, frac=0.01, seed=11, tol=RGBA_TOL):
    """Where the whole-frame checks above still BORROW from the producer: they composite in the producer's sorted order and
    tile ranges.  Here a random sample of tiles (>= `frac` of the populated ones) is composited from the twin's OWN lists:
    the members of a tile are the splats whose own float64 tile rectangle (p["rect"], gsplat_projection.glsl:144-148) covers
    it, ordered by the twin's own depth code (p["depth16"], :218) with ties in ascending splat id — nothing of the
    producer's sort, emission or tile ranges enters.  The records are `own_records` (the caller's choice: the twin's float64
    records, or those with the frame's binary32 centres).  Reported per tile class — the producer's list is IDENTICAL to the
    twin's, the same SET in another order (depth codes that floor differently: 11.7 % of the codes sit within +-8 ulp of a
    floor), or a different set (a rectangle / cull decision the twin calls unstable) — how many pixels differ from the
    producer's frame by more than `tol`, and how many of those sit on a knife edge the twin finds by itself."""
    gx, gy = p["gx"], p["gy"]
    b = np.asarray(bounds, np.int64)
    lens = np.maximum(b[:, 1] - b[:, 0], 0)
    populated = np.flatnonzero(lens > 0)
    # (the frame's highest populated tile carries quirk Q5/Q6 — an empty or shortened range — which is the boundaries
    # shader's doing, not the order's: checked elsewhere, not sampled)
    populated = populated[:-1] if populated.size > 1 else populated
    rng = np.random.default_rng(seed)
    k = max(8, int(np.ceil(frac * populated.size)))
    sample = np.sort(rng.choice(populated, size=min(k, populated.size), replace=False))
    alive = p["alive"]
    rect = p["rect"]
    depth = p["depth16"].astype(np.int64)
    raster = np.asarray(own_records, np.float64)
    classes = {"identical_list": [0, 0, 0, 0.0], "same_set_other_order": [0, 0, 0, 0.0], "different_set": [0, 0, 0, 0.0]}
    flips_total, pixels_total = 0, 0
    examples = []
    for tid in sample:
        tx, ty = int(tid % gx), int(tid // gx)
        member = alive & (rect[:, 0] <= tx) & (tx < rect[:, 2]) & (rect[:, 1] <= ty) & (ty < rect[:, 3])
        ids = np.flatnonzero(member)
        order = np.lexsort((ids, depth[ids]))           # depth code, then splat id
        own_list = ids[order].astype(np.int64)
        got_list = np.asarray(sorted_values[b[tid, 0]:b[tid, 1]], np.int64)
        if own_list.size == got_list.size and np.array_equal(own_list, got_list):
            cls = "identical_list"
        elif own_list.size == got_list.size and np.array_equal(np.sort(own_list), np.sort(got_list)):
            cls = "same_set_other_order"
            flips_total += int((own_list != got_list).sum())
        else:
            cls = "different_set"
        tb = np.zeros((gx * gy, 2), np.int64)
        tb[tid] = (0, own_list.size)
        kw = dict(tiles=(tx, tx + 1, ty, ty + 1))
        base = twin.render(raster, own_list, tb, w, h, **kw)
        lo = twin.render(raster, own_list, tb, w, h, alpha_scale=1 - 4e-6, **kw)
        hi = twin.render(raster, own_list, tb, w, h, alpha_scale=1 + 4e-6, **kw)
        sl = (slice(ty * 16, min(ty * 16 + 16, h)), slice(tx * 16, min(tx * 16 + 16, w)))
        err = np.max(np.abs(np.asarray(img[sl], np.float64) - base[sl]), axis=-1)
        knife = np.max(np.abs(hi[sl] - lo[sl]), axis=-1) > 2e-5
        over = err > tol
        c = classes[cls]
        c[0] += 1
        c[1] += int(over.sum())
        c[2] += int((over & knife).sum())
        c[3] = max(c[3], float(err[~knife].max(initial=0.0)))
        pixels_total += int(err.size)
        if over.any() and len(examples) < 6:
            examples.append({"tile": int(tid), "class": cls, "pairs": int(got_list.size), "pixels_over_tol": int(over.sum()),
                             "of_them_on_knife_edges": int((over & knife).sum()), "max_err": float(err.max()),
                             "list_positions_that_differ": int((own_list != got_list).sum()) if own_list.size == got_list.size else None})
    return {"tiles_sampled": int(sample.size), "of_populated_tiles": int(populated.size), "pixels": pixels_total,
            "order": "the twin's own float64 depth codes, ties in ascending splat id; members from the twin's own rectangles",
            "per_class": {k: {"tiles": v[0], "pixels_over_tol": v[1], "of_them_on_knife_edges": v[2],
                              "max_err_off_knife_edges": v[3]} for k, v in classes.items()},
            "list_positions_in_another_order": flips_total, "examples": examples}
