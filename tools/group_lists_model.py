#!/usr/bin/env python3
"""CPU model of a compositor whose waves walk one work list per 16-lane group (a 4x4 or 8x2 pixel block of the wave's
8x8 quadrant) instead of one per wave: how many blend-loop steps would be left.

    python tools/group_lists_model.py [config] [tiles] [seed]

Uses the oracle (test infrastructure) for the frame's lists and records, then replays a sample of interior tiles in
numpy f32 with the compositor's own rules: batches of 256 staged splats, a pixel leaves at t <= 1/255, a splat is
ignored by a pixel below y = -32, the conservative reach mask taken as "some pixel of the block has y >= -34", the block
early exit of gsplat_render.glsl:66.  Counts, per organisation, the steps that run the whole update ("seen": some alive
lane above the cutoff, 54 cycles measured) and the steps that end at the cutoff test (17.6 cycles)
(profiles/r03_step_rates.md).  Not part of the product; nothing here is timed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from godotgaussiansplatting_amd import capi, scenes  # noqa: E402
from oracle import cpu  # noqa: E402

LOG2E = np.float32(1.4426950408889634)
SEEN, CUT = 54.0, 17.6


def tile_steps(rec, px, py, groupings):
    """rec: (n, 9) f32 = ipx ipy hx hy hz r g b opacity in list order.  Returns {name: [seen, cut]} and staged count."""
    n = rec.shape[0]
    t = np.ones(256, np.float32)
    out = {k: [0, 0, 0] for k in groupings}
    staged = 0
    for off in range(0, n, 256):
        b = rec[off:off + 256]
        m = b.shape[0]
        staged += m
        dx = b[:, 0:1] - px[None, :]
        dy = b[:, 1:2] - py[None, :]
        a1 = b[:, 3:4] * dy + b[:, 2:3] * dx
        y = a1 * dx + (b[:, 4:5] * dy) * dy            # (m, 256)
        above = y >= np.float32(-32.0)
        reach = ~(y < np.float32(-34.0))               # NaN: kept
        e = np.exp2(np.minimum(y, np.float32(126.0)).astype(np.float32))
        alpha = b[:, 8:9] * e
        alive_before = np.empty((m + 1, 256), bool)
        for j in range(m):                              # the only sequential part: transmittance
            alive = t > np.float32(1.0 / 255.0)
            alive_before[j] = alive
            upd = alive & above[j]
            w = alpha[j] * t
            t = np.where(upd, t - w, t)
        alive_before[m] = t > np.float32(1.0 / 255.0)
        act = alive_before[:m] & above                  # lanes that take the update
        for name, groups_of_wave in groupings.items():
            seen = cut = entries = 0
            for groups in groups_of_wave:               # one wave
                lists = [np.flatnonzero(reach[:, g].any(1)) for g in groups]
                steps = max(len(l) for l in lists)
                entries += sum(len(l) for l in lists)
                if steps == 0:
                    continue
                # state of the wave before step k: every group's pixels before its k-th entry (or after its last one)
                full = np.zeros(steps, bool)
                wave_alive = np.zeros(steps, bool)
                for g, l in zip(groups, lists):
                    k = len(l)
                    full[:k] |= act[l][:, g].any(1)
                    wave_alive[:k] |= alive_before[l][:, g].any(1)
                    if k < steps:   # after its last entry nothing in this batch touches the group's pixels
                        wave_alive[k:] |= alive_before[m][g].any()
                dead = np.flatnonzero(~wave_alive)
                stop = dead[0] if dead.size else steps
                seen += int(full[:stop].sum())
                cut += int(stop - full[:stop].sum())
            out[name][0] += seen
            out[name][1] += cut
            out[name][2] += entries
        if m == 256 and int(np.floor(t * np.float32(255.0)).astype(np.int64).sum()) <= 255:
            break
    return out, staged


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    n, deg, w, h, _ = scenes.CONFIGS[cfg]
    cam = scenes.default_camera()
    vp, cam_pos = capi.make_view_proj(cam.xform12(), cam.fov, w / h, cam.near, cam.far)
    rows = scenes.config_rows(cfg)
    records = cpu.records_from_ply_rows(rows)
    frame = cpu.Frame.make(vp, cam_pos, w, h)
    cap = int(os.environ.get("MODEL_CAPACITY", str(12 * n)))
    fr = cpu.render_frame(records, frame, capacity=cap, want_image=False)
    culled, values, bounds = fr["culled"], fr["values"], fr["bounds"].astype(np.int64)
    gx, gy = cpu.grid(w, h)
    lx, ly = np.meshgrid(np.arange(16), np.arange(16))
    lx, ly = lx.ravel(), ly.ravel()                     # pixel p = ly * 16 + lx
    quad = (lx // 8) + 2 * (ly // 8)

    def groups(kind):
        res = []
        for q in range(4):
            inq = np.flatnonzero(quad == q)
            if kind == "wave":
                res.append([inq])
            elif kind == "4x4":
                key = ((lx[inq] % 8) // 4) + 2 * ((ly[inq] % 8) // 4)
                res.append([inq[key == k] for k in range(4)])
            elif kind == "8x2":
                key = (ly[inq] % 8) // 2
                res.append([inq[key == k] for k in range(4)])
            elif kind == "2x2x16":                      # 16 lists per wave of 2x2 pixels: the limit of the idea
                key = ((lx[inq] % 8) // 2) + 4 * ((ly[inq] % 8) // 2)
                res.append([inq[key == k] for k in range(16)])
        return res

    groupings = {k: groups(k) for k in ("wave", "4x4", "8x2")}
    rng = np.random.default_rng(seed)
    interior = [(bx, by) for by in range(gy - 1) for bx in range(gx - 1)]
    pick = rng.choice(len(interior), size=min(n_tiles, len(interior)), replace=False)
    tot = {k: np.zeros(3) for k in groupings}
    staged_tot = 0
    for i in pick:
        bx, by = interior[i]
        b0, b1 = bounds[by * gx + bx]
        if b1 <= b0:
            continue
        ids = values[b0:b1]
        c = culled[ids]
        rec = np.empty((ids.size, 9), np.float32)
        rec[:, 0:2] = c[:, 0:2]
        rec[:, 2] = (np.float32(-0.5) * c[:, 4]) * LOG2E
        rec[:, 3] = (-c[:, 5]) * LOG2E
        rec[:, 4] = (np.float32(-0.5) * c[:, 6]) * LOG2E
        rec[:, 5:9] = c[:, 8:12]
        px = (bx * 16 + lx).astype(np.float32)
        py = (by * 16 + ly).astype(np.float32)
        with np.errstate(all="ignore"):
            o, staged = tile_steps(rec, px, py, groupings)
        staged_tot += staged
        for k in groupings:
            tot[k] += np.array(o[k], float)
    print(f"{cfg}: {len(pick)} interior tiles, {staged_tot} staged pairs "
          f"({staged_tot / len(pick):.0f} per tile; whole frame D = {fr['D']})")
    base = tot["wave"][0] * SEEN + tot["wave"][1] * CUT
    for k, v in tot.items():
        cyc = v[0] * SEEN + v[1] * CUT
        print(f"  {k:>5}: seen {v[0] / staged_tot:.3f}  cut {v[1] / staged_tot:.3f} wave-steps per staged pair, "
              f"list entries {v[2] / staged_tot:.2f} per staged pair, loop cycles {cyc / base:.3f} of today's")


if __name__ == "__main__":
    main()
