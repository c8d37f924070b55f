"""Executable model of the build's TWO-ROUND frame schedule — test infrastructure, like the rest of
oracle/: nothing in the product imports it.

The reference composites every tile from its complete, sorted list (gsplat_render.glsl:61-98) and leaves a tile at the
first 256-pair batch boundary where sum(uint(t * 255)) over its 256 pixels is <= 255 (:66,97).  The HIP build may split
a frame in two: round A emits, sorts and composites the first v_a entries of the (depth16, id)-sorted splat list, round
B emits the remaining splats only where their rectangle still holds an unfinished tile, and unfinished tiles resume from
their per-pixel state with the batch boundaries of the tile's WHOLE list.  This module restates that schedule in NumPy —
emission, the tile-range pass with quirks Q5/Q6 (gsplat_boundaries.glsl:39-49) in its "frame's last tile" form, the
batch / early-exit rule, the treatment of tile T - 1 — on top of the oracle's projection output, with one simple
float32 blend per (pixel, splat) that both schedules share.  tests/test_rounds_model.py checks that the two schedules
produce the same pixels and the same staged counts for any split, which is the claim the GPU tests then check against
the real kernels.  (The blend is NOT the arithmetic contract of DESIGN.md §3: only the schedule is modelled here.)"""
import numpy as np

TILE = 16
BATCH = 256


def splat_list(proj):
    """The visible splats in (depth16, id) order, each with its tiles in emission order (y outer, x inner)."""
    keys, values = proj["keys"], proj["values"]          # emission in ascending id (oracle order)
    ids, first, cnt = np.unique(values, return_index=True, return_counts=True)
    depth = (keys[first] & 0xFFFF).astype(np.int64)
    order = np.lexsort((ids, depth))
    tiles = [(keys[first[k]:first[k] + cnt[k]] >> 16).astype(np.int64) for k in order]
    return ids[order].astype(np.int64), depth[order], tiles


def emit(entries):
    """entries: iterable of (id, depth16, tiles) in list order -> pairs sorted by tile, stable (= by tile, depth16, id)."""
    t, v = [], []
    for sid, _d, tl in entries:
        t.append(tl)
        v.append(np.full(tl.size, sid, np.int64))
    if not t:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    t, v = np.concatenate(t), np.concatenate(v)
    o = np.argsort(t, kind="stable")
    return t[o], v[o]


def tile_ranges(tiles_sorted, num_tiles, frame_last_plus1, fix_last=False):
    """gsplat_boundaries.glsl:23-50 on one round's sorted array; the quirk test asks for the FRAME's highest populated
    tile (for a whole-frame array that is the array's own last tile: the reference's test)."""
    b = np.zeros((num_tiles, 2), np.int64)
    n = tiles_sorted.size
    for i in range(1, n):
        if tiles_sorted[i] != tiles_sorted[i - 1]:
            b[tiles_sorted[i - 1], 1] = i
            b[tiles_sorted[i], 0] = i
    if n:
        cur = int(tiles_sorted[-1])
        if fix_last or cur + 1 != frame_last_plus1:
            b[cur, 1] = n
        elif n > 1 and cur == num_tiles - 1:
            b[cur, 1] = n - 1          # Q6
        # else Q5: the frame's highest populated tile keeps .y = 0
    return b


class TileState:
    def __init__(self):
        self.c = np.zeros((BATCH, 3), np.float32)
        self.t = np.ones(BATCH, np.float32)
        self.consumed = 0
        self.done = False


def composite(state, tile_id, gx, splat_ids, culled):
    """Continue one tile with the next pairs of its list; batches end where the tile's whole list reaches a multiple of
    256 pairs.  Returns True if the tile left at a batch boundary (nothing behind it is ever read)."""
    bx, by = tile_id % gx, tile_id // gx
    lx, ly = np.meshgrid(np.arange(TILE), np.arange(TILE))
    px = (bx * TILE + lx).astype(np.float32).ravel()
    py = (by * TILE + ly).astype(np.float32).ravel()
    off, n = 0, len(splat_ids)
    while off < n:
        chunk = min(BATCH - (state.consumed % BATCH), n - off)
        for sid in splat_ids[off:off + chunk]:
            r = culled[sid]
            dx, dy = r[0] - px, r[1] - py
            power = np.float32(-0.5) * (r[4] * dx * dx + r[6] * dy * dy) - r[5] * dx * dy
            alpha = np.minimum(np.float32(0.99), r[11] * np.exp(np.minimum(power, np.float32(0.0))))
            live = (state.t > np.float32(1.0 / 255.0)) & (power <= 0) & (alpha >= np.float32(1.0 / 255.0))
            w = np.where(live, alpha * state.t, np.float32(0.0)).astype(np.float32)
            state.c += r[8:11][None, :] * w[:, None]
            state.t = (state.t - w).astype(np.float32)
        off += chunk
        state.consumed += chunk
        if state.consumed % BATCH == 0 and int((state.t * np.float32(255.0)).astype(np.uint32).sum()) <= 255:
            return True
    return False


def _finish(states, gx, gy, width, height):
    img = np.zeros((gy * TILE, gx * TILE, 4), np.float32)
    for tid, st in enumerate(states):
        bx, by = tid % gx, tid // gx
        img[by * TILE:(by + 1) * TILE, bx * TILE:(bx + 1) * TILE, :3] = st.c.reshape(TILE, TILE, 3)
        img[by * TILE:(by + 1) * TILE, bx * TILE:(bx + 1) * TILE, 3] = 1.0
    return img[:height, :width]


def one_round(proj, gx, gy, width, height, fix_last=False):
    ids, depth, tiles = splat_list(proj)
    T = gx * gy
    ts, vs = emit(zip(ids, depth, tiles))
    b = tile_ranges(ts, T, proj["frame_last_tile_plus1"], fix_last)
    states = [TileState() for _ in range(T)]
    for tid in range(T):
        x, y = b[tid]
        composite(states[tid], tid, gx, vs[x:max(x, y)], proj["culled"])
    return _finish(states, gx, gy, width, height), np.array([s.consumed for s in states]), ts.size


def _park(state, tile_id, gx, width, height, drop_offimage):
    """The hand-over of an unfinished tile from round A to round B as the kernel does it (raster.hip): colour and
    transmittance of the pixels inside the image wait in the image, the transmittance of the out-of-image lanes of an
    edge tile (they take part in the early-exit sum) in a side buffer.  drop_offimage=True is the defect of
    the round-2 build, kept so the tests can show they see it: those lanes restart round B at t = 1."""
    bx, by = tile_id % gx, tile_id // gx
    lx, ly = np.meshgrid(np.arange(TILE), np.arange(TILE))
    inside = ((bx * TILE + lx < width) & (by * TILE + ly < height)).ravel()
    resumed = TileState()
    resumed.consumed, resumed.done = state.consumed, state.done
    resumed.c[inside] = state.c[inside]
    resumed.t[inside] = state.t[inside]
    if not drop_offimage:
        resumed.t[~inside] = state.t[~inside]
    return resumed


def two_rounds(proj, gx, gy, width, height, frac, fix_last=False, drop_offimage=False):
    ids, depth, tiles = splat_list(proj)
    T = gx * gy
    last = T - 1
    flp1 = proj["frame_last_tile_plus1"]
    v_a = int(frac * ids.size)
    entries = list(zip(ids, depth, tiles))
    # round A
    ts, vs = emit(entries[:v_a])
    b = tile_ranges(ts, T, flp1, fix_last)
    states = [TileState() for _ in range(T)]
    for tid in range(T):
        x, y = b[tid]
        states[tid].done = composite(states[tid], tid, gx, vs[x:max(x, y)], proj["culled"])
    if not states[last].done:          # T - 1 undecided: round B starts it over, from its complete list
        states[last] = TileState()
    states = [st if st.done else _park(st, tid, gx, width, height, drop_offimage) for tid, st in enumerate(states)]
    pairs_a = ts.size
    # round B: A's splats give their T - 1 pair again if that tile is redone; the others emit iff a tile of theirs is unfinished
    done = np.array([s.done for s in states])
    rest = []
    for k, (sid, d, tl) in enumerate(entries):
        if k < v_a:
            if not done[last] and (tl == last).any():
                rest.append((sid, d, np.array([last], np.int64)))
        elif (~done[tl]).any():
            rest.append((sid, d, tl))
    ts, vs = emit(rest)
    b = tile_ranges(ts, T, flp1, fix_last)
    for tid in range(T):
        if states[tid].done:
            continue
        x, y = b[tid]
        composite(states[tid], tid, gx, vs[x:max(x, y)], proj["culled"])
    return _finish(states, gx, gy, width, height), np.array([s.consumed for s in states]), pairs_a + ts.size
