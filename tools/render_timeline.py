"""Where a wave of the compositor spends its time.  Needs a probe build of the library (raster.hip compiled with
-DGS_PROBE_TIMELINE: every wave overwrites two pixels of its quadrant with {start tick, duration, HW_ID, cycles inside
the blend loop | list entries} and {staging, barrier after staging, list build, sum + barriers around the batch});
GSPLAT_LIB must point at it.  (s_memtime is not comparable between XCDs, and not reliably between the SEs of one:
only differences taken inside one wave are used.)

    GSPLAT_LIB=$PWD/build_variants/libgsplat_tl.so GSPLAT_ROUNDS=off python tools/render_timeline.py c3
"""
import sys
sys.path.insert(0, '.')
import numpy as np
from godotgaussiansplatting_amd import capi, scenes
import bench


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
    ctx = capi.Context(n, w, h)
    rows = scenes.config_rows(cfg)
    for first in range(0, n, 1 << 20):
        ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
    fr = capi.make_frame(vp, cam)
    for _ in range(4):  # the schedule of a frame uses the staged counts of the one before; colour mode settles
        ctx.render(fr)
        ctx.synchronize()
    img = ctx.render_to_host(fr)
    staged = ctx.read_tile_staged().astype(np.int64)
    gx, gy = (w + 15) // 16, (h + 15) // 16
    raw = img.view(np.uint32)
    rows = []
    for ty in range(gy):
        for tx in range(gx):
            for q in range(4):
                px, py = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
                if px >= w or py >= h:
                    continue
                t0, dur, hw, bl = [int(v) for v in raw[py, px]]
                ph = [int(v) for v in raw[py, px + 1]] if px + 1 < w else [0, 0, 0, 0]
                rows.append((ty * gx + tx, q, t0 | ((hw >> 24) << 32), dur, hw & 0xFFFF, (hw >> 16) & 0xF, bl & 0xFFFFF, bl >> 20, *ph))
    a = np.array(rows, dtype=np.int64)
    tile, q, t0, dur, hw, xcc, blend, entries, p_stage, p_bar2, p_list, p_tail = a.T
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 1
    simd = (hw >> 4) & 3
    print(f"{cfg}: {len(a)} waves of {gx * gy} tiles")
    print(f"wave duration ticks: mean {dur.mean():.0f} p50 {np.percentile(dur, 50):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max()}; "
          f"inside the blend loop {blend.sum() / dur.sum():.2%} of wave time; list entries per wave mean {entries.mean():.0f}")
    tot = dur.sum()
    print(f"a wave's time: staging (loads + record + colour) {p_stage.sum() / tot:.2%}, barrier after staging {p_bar2.sum() / tot:.2%}, "
          f"list build {p_list.sum() / tot:.2%}, blend loop {blend.sum() / tot:.2%}, t sum + barriers around the batch {p_tail.sum() / tot:.2%}, "
          f"rest (prologue, image write) {1 - (p_stage.sum() + p_bar2.sum() + p_list.sum() + blend.sum() + p_tail.sum()) / tot:.2%}")
    # per-tile: duration of the slowest wave vs staged
    per_tile = {}
    for t, d in zip(tile, dur):
        per_tile[int(t)] = max(per_tile.get(int(t), 0), int(d))
    tt = np.array(sorted(per_tile))
    dd = np.array([per_tile[int(t)] for t in tt])
    st = staged[tt]
    for lo, hi in [(0, 1), (1, 257), (257, 513), (513, 1025), (1025, 2049), (2049, 1 << 30)]:
        mm = (st >= lo) & (st < hi)
        if mm.any():
            print(f"tiles with staged in [{lo}, {hi}): {mm.sum()} tiles, mean duration {dd[mm].mean():.0f} ticks, share of tile time {dd[mm].sum() / dd.sum():.2%}")
    # imbalance inside a workgroup: blend cycles of the four waves
    bl = {}
    for t, b in zip(tile, blend):
        bl.setdefault(int(t), []).append(int(b))
    mx = np.array([max(v) for v in bl.values()], dtype=np.float64)
    mean = np.array([sum(v) / len(v) for v in bl.values()])
    print(f"blend cycles per tile: sum of max over waves / sum of mean = {mx.sum() / max(mean.sum(), 1):.2f}")


if __name__ == '__main__':
    main()
