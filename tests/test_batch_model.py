"""Frames in batches — the construction (B frames of one context as ONE frame of a virtual image that stacks the B stripes;
DESIGN.md §6, csrc/gsplat_internal.h: FrameBatch) as a NumPy model over the oracle's per-frame projections: the batch's one
sorted array holds every frame's sorted pairs, frame after frame, and the batch form of the tile-range pass gives every
frame the tile ranges the oracle computes for it alone — quirk Q5/Q6 (gsplat_boundaries.glsl:39-49) included, which hits a
frame's own highest populated tile in the MIDDLE of the batch's array.  CPU only; the kernels themselves are checked on the
GPU (tests/test_gpu_parity.py::test_batched_frames_*)."""
import numpy as np
import pytest

import oracle
from oracle import batch_model as bm
from conftest import godot_perspective, make_case


def _frames(case, count, w, h, empty=None):
    from godotgaussiansplatting_amd import scenes
    out = []
    for k in range(count):
        ang = 0.3 + 0.45 * k
        cam = scenes.look_at_camera((5.0 * np.sin(ang), 0.5 * np.cos(1.3 * ang), 5.0 * np.cos(ang)))
        if empty == k:   # a frame that sees nothing: the camera looks away from the scene
            cam = scenes.look_at_camera((0.0, 0.0, 60.0), target=(0.0, 0.0, 120.0))
        vp = oracle.pack_camera(cam.xform12(), godot_perspective(cam.fov, w / h, cam.near, cam.far))
        pos = np.array([-cam.origin[0], -cam.origin[1], cam.origin[2]], np.float32)
        out.append((vp, pos))
    return out


@pytest.mark.parametrize("stripe_name,empty", [("full", None), ("rows", None), ("rows-last", None), ("columns", None),
                                                ("full", 1), ("rows", 0)])
def test_a_batch_holds_every_frames_sorted_pairs_and_tile_ranges(stripe_name, empty):
    n, w, h, B = 5000, 160, 112, 3
    gx, gy = oracle.grid(w, h)
    stripe = {"full": (0, gx, 0, gy), "rows": (0, gx, 2, 5), "rows-last": (0, gx, 4, gy), "columns": (3, 8, 0, gy)}[stripe_name]
    case = make_case(n, w, h, seed=501, sh_degree=0, scale_n=1500)
    n_pad = ((n + 511) // 512) * 512
    cams = _frames(case, B, w, h, empty)
    proj, refs, last = [], [], []
    for vp, pos in cams:
        whole = oracle.Frame.make(vp, pos, w, h)
        part = oracle.Frame.make(vp, pos, w, h, 1.0, 0.0, 0.0, 0xFFFFFFFF, None if stripe_name == "full" else stripe)
        proj.append(oracle.project(case["records"], part, capacity=400 * n))
        refs.append(oracle.render_frame(case["records"], part, capacity=400 * n, want_image=False))
        # the FRAME's highest populated tile + 1 (what the ranks' all-reduce(MAX) delivers): from the unclamped rectangles
        last.append(oracle.project(case["records"], whole, capacity=400 * n)["frame_last_tile_plus1"])
        assert proj[-1]["frame_last_tile_plus1"] == last[-1]
    if empty is None:
        assert all(r["D"] > 200 for r in refs) and len({r["D"] for r in refs}) == B
    else:
        assert refs[empty]["D"] == 0 and last[empty] == 0
    lid, d16, vid = bm.sort_by_tile(*bm.virtual_pairs(proj, gx, stripe, n_pad))
    assert lid.size == sum(r["D"] for r in refs)
    sharded = stripe_name != "full"
    bounds = bm.boundaries_batch(lid, gx, stripe, B, last, gx * gy, sharded=sharded)
    sx0, sx1, sy0, sy1 = stripe
    sw, rows = sx1 - sx0, sy1 - sy0
    for b, (start, end) in enumerate(bm.frame_slices(lid, stripe, B)):
        ref = refs[b]
        assert end - start == ref["D"]
        # the frame's part of the batch's array IS the frame's own sorted array (keys rebuilt from local id + depth code)
        l = lid[start:end] - b * rows * sw
        real_tile = (sy0 + l // sw) * gx + sx0 + l % sw
        np.testing.assert_array_equal((real_tile << 16) | d16[start:end], ref["keys"].astype(np.int64), err_msg=f"frame {b}")
        np.testing.assert_array_equal(vid[start:end] - b * n_pad, ref["values"].astype(np.int64), err_msg=f"frame {b}")
        # ... and its tile ranges are the frame's own, shifted by where the frame starts in the batch's array
        rb = ref["bounds"].astype(np.int64)
        for ty in range(sy0, sy1):
            for tx in range(sx0, sx1):
                v = ((b * rows + ty - sy0) * gx + tx)
                r = ty * gx + tx
                got = bounds[v]
                # (ranges are indices into the array they were computed on: the frame's own, or the batch's, where the
                # frame starts at `start` — what must agree is every tile's pair count, quirks included, and, where the
                # oracle wrote a start, the start itself)
                if rb[r][0] > 0:
                    assert got[0] == start + rb[r][0], (b, tx, ty, rb[r], got, start)
                num_ref = max(int(rb[r][1]) - int(rb[r][0]), 0)
                num_got = max(int(got[1]) - int(got[0]), 0)
                assert num_got == num_ref, (b, tx, ty, rb[r], got, start)
    # quirks Q5 / Q6 in the MIDDLE of the batch's array: a frame that is not the batch's last ends on its own highest populated
    # tile — .y stays 0 (Q5), or end - 1 if that tile is T - 1 (Q6) — although pairs of a later frame follow it
    if stripe_name == "full" and empty is None:
        slices = bm.frame_slices(lid, stripe, B)
        q5 = q6 = 0
        for b in range(B - 1):
            t_last = last[b] - 1
            v = (b * gy + t_last // gx) * gx + t_last % gx
            if t_last == gx * gy - 1:
                assert bounds[v][1] == slices[b][1] - 1
                q6 += 1
            else:
                assert bounds[v][1] == 0
                q5 += 1
        assert q5 + q6 == B - 1


def test_quirk_q5_inside_a_batch():
    """A scene that leaves the bottom tile rows empty from every camera of the batch: each frame's highest populated tile is
    not T - 1 and never gets its .y (Q5) — in the batch's array too, where another frame's pairs follow."""
    n, w, h, B = 4000, 160, 112, 3
    gx, gy = oracle.grid(w, h)
    case = make_case(n, w, h, seed=502, sh_degree=0, scale_n=20000)
    rec = case["records"]
    cams = _frames(case, B, w, h)
    # drop every splat that reaches the two bottom tile rows in ANY of the frames
    keep = np.ones(n, bool)
    for vp, pos in cams:
        p = oracle.project(rec, oracle.Frame.make(vp, pos, w, h), capacity=400 * n)
        low = np.unique(p["values"][((p["keys"] >> 16) // gx) >= gy - 2])
        keep[low] = False
    rec = np.ascontiguousarray(rec[keep])
    n2 = rec.shape[0]
    n_pad = ((n2 + 511) // 512) * 512
    stripe = (0, gx, 0, gy)
    proj, refs, last = [], [], []
    for vp, pos in cams:
        fr = oracle.Frame.make(vp, pos, w, h)
        proj.append(oracle.project(rec, fr, capacity=400 * n))
        refs.append(oracle.render_frame(rec, fr, capacity=400 * n, want_image=False))
        last.append(proj[-1]["frame_last_tile_plus1"])
        assert 0 < last[-1] < gx * gy and refs[-1]["D"] > 100
    lid, d16, vid = bm.sort_by_tile(*bm.virtual_pairs(proj, gx, stripe, n_pad))
    bounds = bm.boundaries_batch(lid, gx, stripe, B, last, gx * gy, sharded=False)
    for b, (start, end) in enumerate(bm.frame_slices(lid, stripe, B)):
        rb = refs[b]["bounds"].astype(np.int64)
        t_last = last[b] - 1
        assert rb[t_last][1] == 0 and rb[t_last][0] > 0                      # the oracle's own frame: Q5
        v = (b * gy + t_last // gx) * gx + t_last % gx
        assert bounds[v][1] == 0 and bounds[v][0] == start + rb[t_last][0]  # ... and the batch's
        for r in range(gx * gy):
            vv = (b * gy + r // gx) * gx + r % gx
            assert max(int(bounds[vv][1]) - int(bounds[vv][0]), 0) == max(int(rb[r][1]) - int(rb[r][0]), 0)
