"""Independent NumPy twin of the reference pipeline — TEST INFRASTRUCTURE ONLY.

Written from the GLSL text, not from gsplat_oracle.c: every expression is the literal one of the shader
(gsplat_projection.glsl, gsplat_boundaries.glsl, gsplat_render.glsl), evaluated in binary64 with NumPy's
exp/power, i.e. "ideal arithmetic".  It pins the C oracle from a second direction:
  * integer outputs (tile rectangles, depth codes, tile ranges) must agree except where an input sits within
    rounding distance of a discontinuity (a handful of splats per 10^5);
  * RasterizeData must agree to binary32 rounding; the image to ~1e-5 except knife-edge pixels.
Slow (Python loop over tiles): use scenes of a few thousand splats.
"""
import numpy as np

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, 1.0925484305920792, 0.31539156525252005, 1.0925484305920792, 0.5462742152960396]
SH_C3 = [0.5900435899266435, 2.890611442640554, 0.4570457994644658, 0.3731763325901154, 0.4570457994644658,
         1.445305721320277, 0.5900435899266435]


def _f32c(x):
    """GLSL float literals are binary32."""
    return float(np.float32(x))


def project(records, view, proj, cam_pos, model_scale, width, height, time=0.0):
    """gsplat_projection.glsl:150-226 for all splats.  Returns dict of per-splat arrays (float64 / int64)."""
    r = np.asarray(records, np.float64).reshape(-1, 60)
    n = r.shape[0]
    V = np.asarray(view, np.float64).reshape(4, 4).T   # column-major -> math matrix
    P = np.asarray(proj, np.float64).reshape(4, 4).T
    gx, gy = (width + 15) // 16, (height + 15) // 16
    pos = r[:, 0:3] * model_scale                                           # :160
    view_pos = (V @ np.concatenate([pos, np.ones((n, 1))], 1).T).T          # :161
    clip = (P @ view_pos.T).T                                               # :162
    vb = clip[:, 3] * _f32c(1.2)                                            # :163
    culled = ((clip[:, 0] < -vb) | (clip[:, 1] < -vb) | (clip[:, 2] < 0) |
              (clip[:, 0] > vb) | (clip[:, 1] > vb) | (clip[:, 2] > clip[:, 3]))  # :164
    st = time - r[:, 3]                                                     # :169

    def eoc(x):
        a = 1.0 - x
        return 1.0 - a * a * a
    tf = eoc(np.clip(st, 0, 1))
    tfl = eoc(np.clip(st - _f32c(0.35), 0, 1))
    opacity = r[:, 10] * tfl * tfl                                          # :173
    scale = model_scale * (2.0 * (1 - tfl) + 1.0 * tfl)                     # :174 mix(2,1,tfl)
    c = r[:, 4:10] * (scale * scale)[:, None]                               # :125
    cov3 = np.stack([np.stack([c[:, 0], c[:, 1], c[:, 2]], 1), np.stack([c[:, 1], c[:, 3], c[:, 4]], 1),
                     np.stack([c[:, 2], c[:, 4], c[:, 5]], 1)], 1)          # :29
    tan_fov_inv = np.array([P[0, 0], P[1, 1]])                              # :127
    focal = np.array([width, height]) * 0.5 * tan_fov_inv                   # :128
    tan_fov = 1.0 / tan_fov_inv
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        z_inv = 1.0 / view_pos[:, 2]
        f = focal[None, :] * z_inv[:, None]                                 # :131
        m = np.clip(view_pos[:, 0:2] * z_inv[:, None], -tan_fov * _f32c(1.3), tan_fov * _f32c(1.3))  # :133
        J = np.zeros((n, 3, 3))                                             # math matrix (row, col), :134-137
        J[:, 0, 0] = f[:, 0]
        J[:, 1, 1] = f[:, 1]
        J[:, 2, 0] = -f[:, 1] * m[:, 0]                                     # focal.y (sic)
        J[:, 2, 1] = -f[:, 1] * m[:, 1]
        inv_view = V[:3, :3].T                                              # :138 transpose(mat3(view))
        B = inv_view[None] @ J                                              # :139
        cov2 = np.transpose(B, (0, 2, 1)) @ cov3 @ B                        # :140
        a = cov2[:, 0, 0] + _f32c(0.3)
        b = cov2[:, 1, 0]                                                   # cov_2d[0][1] = column 0, row 1
        cc = cov2[:, 1, 1] + _f32c(0.3)
        det = a * cc - b * b                                                # :177
        mid = 0.5 * (a + cc)
        disc = np.sqrt(np.maximum(_f32c(0.1), mid * mid - det))             # :181
        l1, l2 = mid + disc, mid - disc
        alive = ~culled & (det != 0) & ~(l1 < 0) & ~(l2 < 0)
        ndc = clip[:, 0:3] / clip[:, 3:4]                                   # :184
        ipx = ((ndc[:, 0] + 1) * 0.5 - 1.0 * (1 - tf)) * (width - 1)        # :185
        ipy = ((ndc[:, 1] + 1) * 0.5 - 0.75 * (1 - tf)) * (height - 1)
        radius = np.power(np.maximum(opacity, 0), 0.2) * 2.5 * np.sqrt(np.maximum(l1, l2))  # :190
        x0 = np.trunc(np.clip((ipx - radius) / 16, 0, gx))                  # :146
        y0 = np.trunc(np.clip((ipy - radius) / 16, 0, gy))
        x1 = np.clip(np.ceil((ipx + radius) / 16), 0, gx)                   # :147
        y1 = np.clip(np.ceil((ipy + radius) / 16), 0, gy)
        count = np.where(alive, (x1 - x0) * (y1 - y0), 0)
        alive &= count > 0
        d = pos - np.asarray(cam_pos, np.float64)[None, :]                  # :198
        vd = d / np.linalg.norm(d, axis=1, keepdims=True)
        x, y, z = vd[:, 0], vd[:, 1], vd[:, 2]
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        sh = r[:, 12:60].reshape(n, 16, 3)
        C = lambda i: sh[:, i, :]                                           # noqa: E731
        k = lambda v: _f32c(v)                                              # noqa: E731
        col = (0.5 + C(0) * k(SH_C0) - C(1) * k(SH_C1) * y[:, None] + C(2) * k(SH_C1) * z[:, None]
               - C(3) * k(SH_C1) * x[:, None]
               + C(4) * k(SH_C2[0]) * xy[:, None] - C(5) * k(SH_C2[1]) * yz[:, None]
               + C(6) * k(SH_C2[2]) * (2 * zz - xx - yy)[:, None] - C(7) * k(SH_C2[3]) * xz[:, None]
               + C(8) * k(SH_C2[4]) * (xx - yy)[:, None]
               - C(9) * k(SH_C3[0]) * (y * (3 * xx - yy))[:, None] + C(10) * k(SH_C3[1]) * (x * yz)[:, None]
               - C(11) * k(SH_C3[2]) * (y * (4 * zz - xx - yy))[:, None]
               + C(12) * k(SH_C3[3]) * (z * (2 * zz - 3 * xx - 3 * yy))[:, None]
               - C(13) * k(SH_C3[4]) * (x * (4 * zz - xx - yy))[:, None]
               + C(14) * k(SH_C3[5]) * (z * (xx - yy))[:, None] - C(15) * k(SH_C3[6]) * (x * (xx - 3 * yy))[:, None])
        col = np.maximum(0, col)                                            # :100
        conic = np.stack([cc, -b, a], 1) / det[:, None]                     # :202
        depth = np.floor(ndc[:, 2] ** 3 * 65535.0)                          # :218
    raster = np.zeros((n, 12))
    raster[:, 0], raster[:, 1] = ipx, ipy
    raster[:, 2], raster[:, 3] = pos[:, 0], pos[:, 1]
    raster[:, 4:7] = conic
    raster[:, 7] = pos[:, 2]
    raster[:, 8:11] = col
    raster[:, 11] = opacity
    rect = np.stack([x0, y0, x1, y1], 1)
    # Which of the integer decisions would ANY binary32 evaluation of the same expressions take?  Decided here, by the
    # twin alone: a decision is "stable" when it stays the same with its real-valued inputs moved by `tol` relative
    # (a few dozen binary32 ulps, far more than a float evaluation of these short expressions can be off).
    tol = 2.0 ** -18
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        def span(v):
            return tol * np.maximum(1.0, np.abs(v))
        stable_rect = np.ones(n, bool)
        for ip, lim, rad_sign in ((ipx, gx, -1), (ipy, gy, -1), (ipx, gx, +1), (ipy, gy, +1)):
            lo = (ip + rad_sign * radius - span(ip) - span(radius)) / 16
            hi = (ip + rad_sign * radius + span(ip) + span(radius)) / 16
            if rad_sign < 0:
                stable_rect &= np.trunc(np.clip(lo, 0, lim)) == np.trunc(np.clip(hi, 0, lim))
            else:
                stable_rect &= np.clip(np.ceil(lo), 0, lim) == np.clip(np.ceil(hi), 0, lim)
        w = clip[:, 3]
        margins = np.stack([clip[:, 0] + vb, clip[:, 1] + vb, clip[:, 2], vb - clip[:, 0], vb - clip[:, 1],
                            w - clip[:, 2]], 1)
        scale_c = np.maximum(1e-30, np.abs(clip).max(axis=1) * 1.2)
        stable_alive = (np.abs(margins).min(axis=1) > 64 * tol * scale_c) & (np.abs(det) > 0) \
            & (np.abs(l1) > 64 * tol * np.maximum(1.0, np.abs(mid))) & (np.abs(l2) > 64 * tol * np.maximum(1.0, np.abs(mid)))
        dcode = ndc[:, 2] ** 3 * 65535.0
        stable_depth = np.floor(dcode * (1 - tol / 4)) == np.floor(dcode * (1 + tol / 4))   # +-8 binary32 ulps
    return {"alive": alive, "raster": raster, "rect": np.where(alive[:, None], rect, 0).astype(np.int64),
            "count": count.astype(np.int64) * alive, "depth16": np.where(alive, depth, 0).astype(np.int64) & 0xFFFF,
            "stable_rect": stable_rect, "stable_alive": stable_alive, "stable_depth": stable_depth,
            "gx": gx, "gy": gy}


def emit_and_sort(p):
    """:218-226 duplication in ascending splat id (deterministic member), then the stable sort contract."""
    keys, vals = [], []
    gx = p["gx"]
    for i in np.nonzero(p["alive"])[0]:
        x0, y0, x1, y1 = p["rect"][i]
        for ty in range(y0, y1):
            for tx in range(x0, x1):
                keys.append(((ty * gx + tx) << 16) | int(p["depth16"][i]))
                vals.append(i)
    keys = np.array(keys, np.uint32)
    vals = np.array(vals, np.uint32)
    order = np.argsort(keys, kind="stable")
    return keys, vals, keys[order], vals[order]


def boundaries(sorted_keys, num_tiles):
    """gsplat_boundaries.glsl:23-50 executed thread by thread (any order gives the same result)."""
    b = np.zeros((num_tiles, 2), np.uint32)
    d = len(sorted_keys)
    t = sorted_keys >> 16
    for i in range(1, d):
        if t[i - 1] != t[i]:
            b[t[i - 1], 1] = i
            b[t[i], 0] = i
        if t[i] == num_tiles - 1:
            b[num_tiles - 1, 1] = d - 1
    return b


def render(raster, sorted_vals, bounds, width, height, heatmap_factor=0.0, alpha_scale=1.0, tiles=None):
    """gsplat_render.glsl:50-101, one tile at a time, 256 'threads' vectorised; literal expressions in f64.
    alpha_scale != 1 perturbs every alpha by that factor: two runs at 1 -+ a few 1e-6 bracket what any float32 evaluation
    of the same expressions may legitimately produce, and the pixels where they differ by more than rounding noise are
    the ones sitting on a discontinuity (the t <= 1/255 stop, the block early-exit sum) — a knife-edge mask that owes
    nothing to the C oracle.  tiles = (x0, x1, y0, y1): only that rectangle of tiles (the rest of the image stays 0)."""
    gx, gy = (width + 15) // 16, (height + 15) // 16
    tx0, tx1, ty0, ty1 = tiles if tiles is not None else (0, gx, 0, gy)
    img = np.zeros((height, width, 4))
    MIN_ALPHA = 1.0 / 255
    ly, lx = np.divmod(np.arange(256), 16)
    for by in range(ty0, ty1):
        for bx in range(tx0, tx1):
            tid = by * gx + bx
            b0, b1 = int(bounds[tid, 0]), int(bounds[tid, 1])
            num = (b1 - b0) & 0xFFFFFFFF          # :61 uint difference reinterpreted as int, clamped at 0
            num = 0 if num >= 2 ** 31 else num
            iters = -(-num // 256)
            px = (bx * 16 + lx).astype(np.float64)
            py = (by * 16 + ly).astype(np.float64)
            C = np.zeros((256, 3))
            t = np.ones(256)
            shared_t = 0xFFFFFFFF
            i = 0
            while i < iters and shared_t > 255:
                chunk = min(256, num - 256 * i)
                ids = sorted_vals[b0 + 256 * i: b0 + 256 * i + chunk]
                for j in range(chunk):
                    live = t > MIN_ALPHA
                    if not live.any():
                        break
                    rr = raster[ids[j]]
                    dx, dy = rr[0] - px, rr[1] - py
                    power = -0.5 * (rr[4] * dx * dx + rr[6] * dy * dy) - rr[5] * dx * dy   # :84
                    alpha = rr[11] * np.exp(power) * alpha_scale                           # :86
                    C = np.where(live[:, None], C + rr[8:11][None, :] * (alpha * t)[:, None], C)  # :89
                    t = np.where(live, t * (1 - alpha), t)                                 # :90
                shared_t = int(np.sum(np.floor(t * 255).astype(np.int64)))                 # :97
                i += 1
            a = num * _f32c(5e-4)
            heat = (np.array([0, 0, 1.0]) * (1 - a) + np.array([1.0, _f32c(0.2), _f32c(0.2)]) * a)[None, :] \
                * (1 - t)[:, None] * heatmap_factor                                        # :100
            ok = (px < width) & (py < height)
            img[py[ok].astype(int), px[ok].astype(int), :3] = (C + heat)[ok]
            img[py[ok].astype(int), px[ok].astype(int), 3] = 1.0
    return img
