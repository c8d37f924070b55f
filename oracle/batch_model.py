"""Frames in batches (csrc/gsplat_internal.h: FrameBatch) as an executable NumPy model — TEST INFRASTRUCTURE ONLY.

The HIP build renders B frames of one context through one launch sequence as ONE frame of a virtual image that stacks the B
stripes vertically.  This module restates what the batch-aware kernels do to the oracle's per-frame projections — virtual
splat ids, virtual stripe-local tile ids, the one stable sort over all B frames' pairs, and the tile-range pass with quirk
Q5/Q6 (gsplat_boundaries.glsl:39-49) applied at every REAL frame's end inside the batch's sorted array — so that the CPU
suite can check the construction against the oracle's own frames (tests/test_batch_model.py); the GPU suite checks the
kernels themselves (tests/test_gpu_parity.py::test_batched_frames_*)."""
import numpy as np


def virtual_pairs(projections, gx, stripe, n_pad):
    """projections: per frame, oracle.project() of the frame restricted to `stripe` = (sx0, sx1, sy0, sy1) — its pairs in
    the oracle's emission order.  Returns the batch's pairs as the emission kernel writes them: splats in ascending
    (depth16, virtual id) over ALL frames, each splat's tiles y outer / x inner; per pair (local virtual tile id, depth16,
    virtual splat id)."""
    sx0, sx1, sy0, sy1 = stripe
    sw, rows = sx1 - sx0, sy1 - sy0
    lids, depths, vids, seq = [], [], [], []
    for b, p in enumerate(projections):
        k = p["keys"].astype(np.int64)
        tile, d16 = k >> 16, k & 0xFFFF
        tx, ty = tile % gx, tile // gx
        assert ((tx >= sx0) & (tx < sx1) & (ty >= sy0) & (ty < sy1)).all(), "a pair outside the stripe"
        lids.append((b * rows + ty - sy0) * sw + (tx - sx0))          # project_kernel<.., BATCH>: origin row += b * rows - sy0
        depths.append(d16)
        vids.append(b * n_pad + p["values"].astype(np.int64))          # value = frame * n_pad + slot
        seq.append(np.arange(k.size))                                  # a splat's pairs keep their y-outer / x-inner order
    lid, d16, vid, s = (np.concatenate(a) if a else np.zeros(0, np.int64) for a in (lids, depths, vids, seq))
    order = np.lexsort((s, vid, d16))                                  # the splat sort: (depth16, virtual id); pairs in order
    return lid[order], d16[order], vid[order]


def sort_by_tile(lid, d16, vid):
    """The pair level: ONE stable sort on the stripe-local virtual tile id."""
    order = np.argsort(lid, kind="stable")
    return lid[order], d16[order], vid[order]


def boundaries_batch(sorted_lid, gx, stripe, frames, frame_last_tiles, real_tiles, fix_last_tile=False, sharded=True):
    """raster.hip: boundaries_batch_kernel.  Tile ranges at the VIRTUAL tile id (a grid of gx x frames * rows tiles); where
    the array passes from frame f to a later one — and at its end — f's last tile is closed the way a single frame's array
    end is: with f's own "last tile + 1" word and real tile id."""
    sx0, sx1, sy0, sy1 = stripe
    sw, rows = sx1 - sx0, sy1 - sy0
    per_frame = rows * sw
    bounds = np.zeros((gx * frames * rows, 2), np.int64)
    count = sorted_lid.size

    def vglobal(lid):
        return (lid // sw) * gx + sx0 + lid % sw

    def close_frame(lid, end):
        f = lid // per_frame
        vrow = lid // sw
        real = (sy0 + (vrow - f * rows)) * gx + sx0 + lid % sw
        if fix_last_tile or (sharded and real + 1 != frame_last_tiles[f]):
            bounds[vglobal(lid), 1] = end
        elif end > 1 and real == real_tiles - 1:
            bounds[vglobal(lid), 1] = end - 1

    for i in range(count):
        cur = int(sorted_lid[i])
        if i > 0:
            prev = int(sorted_lid[i - 1])
            if prev != cur:
                bounds[vglobal(cur), 0] = i
                if prev // per_frame == cur // per_frame:
                    bounds[vglobal(prev), 1] = i
                else:
                    close_frame(prev, i)
        if i == count - 1:
            close_frame(cur, count)
    return bounds


def frame_slices(sorted_lid, stripe, frames):
    """[start, end) of every frame's part of the batch's sorted array."""
    sx0, sx1, sy0, sy1 = stripe
    per_frame = (sy1 - sy0) * (sx1 - sx0)
    f = sorted_lid // per_frame
    return [(int(np.searchsorted(f, b, "left")), int(np.searchsorted(f, b, "right"))) for b in range(frames)]
