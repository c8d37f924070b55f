"""Known-answer tests that pin the CPU oracle (oracle/gsplat_oracle.c).

The reference ships no tests or golden vectors, so these are hand-derived from the shader text:
closed-form cases, the quirks of SURVEY.md §8a-Q, and the contract math.  CPU only.
"""
import numpy as np
import pytest

import oracle
from conftest import godot_perspective, make_case, oracle_frame


def record(pos=(0, 0, 0), cov=(1, 0, 0, 1, 0, 1), opacity=0.5, dc=(0, 0, 0), load_time=-10.0):
    r = np.zeros(60, np.float32)
    r[0:3] = pos
    r[3] = load_time
    r[4:10] = cov
    r[10] = opacity
    r[12:15] = dc
    return r


def default_vp(w, h):
    from godotgaussiansplatting_amd import scenes
    cam = scenes.default_camera()
    return oracle.pack_camera(cam.xform12(), godot_perspective(cam.fov, w / h, cam.near, cam.far)), \
        np.array([0, 0, 5], np.float32)


# ---------------------------------------------------------------- contract math
def test_pow02_is_the_correctly_rounded_fifth_root():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(200000, dtype=np.float32), np.exp(rng.uniform(-80, 3, 200000)).astype(np.float32)])
    got = oracle.pow02(x)
    want = np.power(x.astype(np.float64), 0.2).astype(np.float32)
    assert np.array_equal(got, want)
    np.testing.assert_array_equal(oracle.pow02(np.float32([0, 1, 32, 243, 1 / 32, -1])), np.float32([0, 1, 2, 3, 0.5, 0]))


def test_exp2_contract():
    ints = np.arange(-125, 127, dtype=np.float32)
    np.testing.assert_array_equal(oracle.exp2(ints), np.exp2(ints.astype(np.float64)).astype(np.float32))  # exact
    x = np.linspace(-30, 4, 2_000_001).astype(np.float32)
    got = oracle.exp2(x).astype(np.float64)
    want = np.exp2(x.astype(np.float64))
    ulp = np.abs(got - want) / (want * 2.0 ** -24)
    assert ulp.max() < 3.0  # inside Vulkan's 3 + 2|x| ulp allowance for exp()
    assert oracle.exp2(np.float32([-1000.0]))[0] == np.float32(2.0 ** -125)  # clamp, never 0/denormal
    assert oracle.exp2(np.float32([500.0]))[0] == np.float32(2.0 ** 126)


# ---------------------------------------------------------------- projection, closed form
def test_single_isotropic_splat_closed_form():
    w, h = 320, 176
    vp, cam = default_vp(w, h)
    s2 = 0.04
    rec = record(cov=(s2, 0, 0, s2, 0, s2), opacity=0.75, dc=(1.0, -0.5, 0.2))[None]
    out = oracle.render_frame(rec, oracle.Frame.make(vp, cam, w, h))
    assert out["stats"]["visible"] == 1
    rd = out["culled"][0]
    # image centre maps with (dims - 1)
    np.testing.assert_allclose(rd[0:2], [(w - 1) / 2, (h - 1) / 2], rtol=1e-6)
    cot = 1 / np.tan(np.radians(75 / 2))
    fy = h * 0.5 * cot / 5.0           # focal.y / distance
    fx = w * 0.5 * (cot / (w / h)) / 5.0
    a, c = fx * fx * s2 + 0.3, fy * fy * s2 + 0.3
    np.testing.assert_allclose(rd[4:7], [1 / a, 0.0, 1 / c], rtol=2e-5, atol=1e-7)  # conic = (c,-b,a)/det
    col = np.maximum(0, 0.5 + 0.28209479177387814 * np.array([1.0, -0.5, 0.2]))
    np.testing.assert_allclose(rd[8:11], col, rtol=1e-6)
    assert rd[11] == np.float32(0.75)
    # depth code: ndc.z = (P22*vz + P32)/(-vz) with vz = -5
    near, far = 0.05, 4000.0
    ndcz = (-(far + near) / (far - near) * -5 + (-2 * near * far / (far - near))) / 5
    assert (out["keys"][0] & 0xFFFF) == int(ndcz ** 3 * 65535)
    # pixels: single splat => C = colour * alpha, alpha = opacity * exp(-0.5 (dx^2/a + dy^2/c))
    img = out["image"]
    for (px, py) in [(159, 87), (160, 88), (163, 90), (150, 80)]:
        dx, dy = (w - 1) / 2 - px, (h - 1) / 2 - py
        alpha = 0.75 * np.exp(-0.5 * (dx * dx / a + dy * dy / c))
        np.testing.assert_allclose(img[py, px, :3], col * alpha, rtol=2e-5, atol=1e-7)
        assert img[py, px, 3] == 1.0
    # radius = opacity^0.2 * 2.5 * sqrt(max eigenvalue): tile rectangle
    radius = 0.75 ** 0.2 * 2.5 * np.sqrt(max(a, c))
    gx = (w + 15) // 16
    x0, x1 = int(((w - 1) / 2 - radius) / 16), int(np.ceil(((w - 1) / 2 + radius) / 16))
    y0, y1 = int(((h - 1) / 2 - radius) / 16), int(np.ceil(((h - 1) / 2 + radius) / 16))
    tiles = sorted(int(k >> 16) for k in out["keys_unsorted"])
    assert tiles == sorted(y * gx + x for y in range(y0, y1) for x in range(x0, x1))


def test_frustum_cull_and_behind_camera():
    w, h = 128, 96
    vp, cam = default_vp(w, h)
    recs = np.stack([record(pos=(0, 0, 0)), record(pos=(0, 0, 10.0)),      # behind the camera (z > 5)
                     record(pos=(40.0, 0, 0)), record(pos=(0, 0, -5000.0))])  # far to the side / beyond far plane
    out = oracle.render_frame(recs, oracle.Frame.make(vp, cam, w, h))
    assert list(out["counts"] > 0) == [True, False, False, False]


def test_load_animation_factors():
    """gsplat_projection.glsl:169-174: opacity * tfl^2, image_pos slides in from (1, 0.75)."""
    w, h = 128, 96
    vp, cam = default_vp(w, h)
    rec = record(opacity=0.8, load_time=0.0)[None]
    for t, tf, tfl in [(0.0, 0.0, 0.0), (0.5, 1 - 0.5 ** 3, 1 - 0.85 ** 3), (2.0, 1.0, 1.0)]:
        out = oracle.project(rec, oracle.Frame.make(vp, cam, w, h, time=t))
        if out["counts"][0] == 0:
            continue
        rd = out["culled"][0]
        np.testing.assert_allclose(rd[11], 0.8 * tfl * tfl, rtol=1e-6)
        np.testing.assert_allclose(rd[0], (0.5 - (1 - tf)) * (w - 1), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(rd[1], (0.5 - 0.75 * (1 - tf)) * (h - 1), rtol=1e-5, atol=1e-4)


# ---------------------------------------------------------------- sort + emission order
def test_sort_is_stable_and_full_key():
    rng = np.random.default_rng(3)
    k = rng.integers(0, 50, 10000, dtype=np.uint32) << rng.integers(0, 27, 10000, dtype=np.uint32)
    v = np.arange(10000, dtype=np.uint32)
    sk, sv = oracle.sort_pairs(k, v)
    order = np.argsort(k, kind="stable")
    np.testing.assert_array_equal(sk, k[order])
    np.testing.assert_array_equal(sv, v[order])


def test_equal_keys_keep_ascending_splat_id():
    w, h = 64, 48
    vp, cam = default_vp(w, h)
    recs = np.stack([record(opacity=0.3, dc=(i, 0, 0)) for i in range(5)])  # identical position => identical keys
    out = oracle.render_frame(recs, oracle.Frame.make(vp, cam, w, h))
    per_tile = {}
    for k, v in zip(out["keys"], out["values"]):
        per_tile.setdefault(int(k), []).append(int(v))
    assert all(ids == sorted(ids) and len(ids) == 5 for ids in per_tile.values())
    # emission: splat-major, y outer / x inner (gsplat_projection.glsl:219-226)
    ku, vu = out["keys_unsorted"], out["values_unsorted"]
    assert np.all(np.diff(vu.astype(np.int64)) >= 0)
    first = ku[vu == 0] >> 16
    assert np.all(np.diff(first.astype(np.int64)) > 0)


# ---------------------------------------------------------------- boundaries quirks
def test_boundaries_quirk_last_populated_tile_has_no_end():
    keys = (np.array([0, 0, 2, 2, 2, 5], np.uint32) << 16) | 7
    b = oracle.boundaries(keys, 8)
    assert b.tolist() == [[0, 2], [0, 0], [2, 5], [0, 0], [0, 0], [5, 0], [0, 0], [0, 0]]  # tile 5: (5,0) => n = 0


def test_boundaries_quirk_last_tile_drops_its_last_splat():
    keys = (np.array([0, 5, 5, 5], np.uint32) << 16) | 7
    b = oracle.boundaries(keys, 6)
    assert b[0].tolist() == [0, 1] and b[5].tolist() == [1, 3]  # D-1 = 3: 2 of the 3 splats survive


def test_boundaries_empty_and_single():
    assert not oracle.boundaries(np.zeros(0, np.uint32), 4).any()
    assert not oracle.boundaries(np.array([3 << 16], np.uint32), 4).any()  # thread 0 returns early


# ---------------------------------------------------------------- compositor rules
def _tile_case(n_splats, opacity, w=16, h=16):
    vp, cam = default_vp(w, h)
    recs = np.stack([record(cov=(4.0, 0, 0, 4.0, 0, 4.0), opacity=opacity, dc=(0.5, 0.5, 0.5))] * n_splats)
    return recs, oracle.Frame.make(vp, cam, w, h)


def test_block_early_exit_skips_later_batches():
    recs, fr = _tile_case(601, 0.999)  # one tile, 600 usable splats (Q6 drops one), every pixel saturates at once
    out = oracle.render_frame(recs, fr)
    assert out["bounds"][0].tolist() == [0, 600]
    assert out["stats"]["composited"] == 256          # only the first LDS batch was fetched
    assert out["stats"]["evals"] < 256 * 64           # each pixel stops after a few splats (t <= 1/255)


def test_no_early_exit_when_pixels_stay_transparent():
    recs, fr = _tile_case(601, 0.001)
    out = oracle.render_frame(recs, fr)
    assert out["stats"]["composited"] == 600
    assert out["stats"]["evals"] == 600 * 256


def test_alpha_is_one_and_background_black():
    vp, cam = default_vp(128, 96)
    recs = np.stack([record(cov=(1e-3, 0, 0, 1e-3, 0, 1e-3), opacity=0.5, dc=(0.5, 0.5, 0.5))] * 3)
    out = oracle.render_frame(recs, oracle.Frame.make(vp, cam, 128, 96))
    assert np.all(out["image"][..., 3] == 1.0)
    assert np.all(out["image"][0, 0, :3] == 0.0)      # tiles no splat touches stay black
    assert out["image"][47, 63, 0] > 0.0              # tile (3,2)
    assert np.all(out["image"][48, 64, :3] == 0.0)    # tile (4,3) is the highest populated tile: black (Q5)


def test_heatmap_tint_formula():
    recs, fr = _tile_case(50, 0.2)
    fr.heatmap_factor = 1.0
    out = oracle.render_frame(recs, fr)
    fr.heatmap_factor = 0.0
    base = oracle.render_frame(recs, fr)
    n = int(out["bounds"][0, 1]) - int(out["bounds"][0, 0])
    a = n * 5e-4
    # t is not exposed; recover (1 - t) from the blue channel and check the other two against it
    dh = out["image"][..., :3].astype(np.float64) - base["image"][..., :3]
    om = dh[..., 2] / (1.0 * (1 - a) + 0.2 * a)
    np.testing.assert_allclose(dh[..., 0], a * om, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dh[..., 1], 0.2 * a * om, rtol=1e-4, atol=1e-6)


def test_pick_ten_percent_rule():
    case = make_case(4000, 128, 96, seed=5)
    gx = 8
    tile = 3 * gx + 4
    out = oracle.render_frame(case["records"], oracle_frame(dict(case, target_tile=tile)))
    b0, b1 = [int(v) for v in out["bounds"][tile]]
    assert b1 > b0
    sid = out["values"][b0 + (b1 - b0) // 10]
    rd = out["culled"][sid]
    np.testing.assert_array_equal(out["pick"], np.float32([rd[2], rd[3], rd[7], b1 - b0]))
    empty = oracle.render_frame(case["records"], oracle_frame(dict(case, target_tile=0xFFFFFFFF)))
    assert not empty["pick"].any()


# ---------------------------------------------------------------- loader + camera
def test_ply_row_transform():
    row = np.zeros(62, np.float32)
    row[0:3] = [1, 2, 3]
    row[6:9] = [0.1, 0.2, 0.3]
    row[9:54] = np.arange(45) + 1           # f_rest: channel-major, 15 per channel
    row[54] = 0.0                           # sigmoid -> 0.5
    row[55:58] = np.log([1.0, 2.0, 3.0])
    row[58:62] = [2.0, 0, 0, 0]             # un-normalised identity quaternion (w first)
    rec = oracle.records_from_ply_rows(row[None], load_time=1.5)[0]
    np.testing.assert_array_equal(rec[0:4], np.float32([1, 2, 3, 1.5]))
    np.testing.assert_allclose(rec[4:10], [1, 0, 0, 4, 0, 9], rtol=1e-6, atol=1e-7)
    assert rec[10] == np.float32(0.5) and rec[11] == 0
    np.testing.assert_array_equal(rec[12:15], np.float32([0.1, 0.2, 0.3]))
    # coefficient k (1..15): (r, g, b) = (f_rest[k-1], f_rest[15+k-1], f_rest[30+k-1])
    np.testing.assert_array_equal(rec[15:18], np.float32([1, 16, 31]))
    np.testing.assert_array_equal(rec[57:60], np.float32([15, 30, 45]))
    # a rotated, anisotropic splat: cov = R diag(s^2) R^T with R from the normalised quaternion
    row[58:62] = [0.9, 0.1, -0.3, 0.2]
    rec = oracle.records_from_ply_rows(row[None])[0]
    q = np.array([0.9, 0.1, -0.3, 0.2])
    q /= np.linalg.norm(q)
    wq, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - wq * z), 2 * (x * z + wq * y)],
                  [2 * (x * y + wq * z), 1 - 2 * (x * x + z * z), 2 * (y * z - wq * x)],
                  [2 * (x * z - wq * y), 2 * (y * z + wq * x), 1 - 2 * (x * x + y * y)]])
    cov = R @ np.diag([1.0, 4.0, 9.0]) @ R.T
    np.testing.assert_allclose(rec[4:10], [cov[0, 0], cov[0, 1], cov[0, 2], cov[1, 1], cov[1, 2], cov[2, 2]],
                               rtol=2e-6, atol=1e-6)


def test_camera_packing_default_camera():
    """gaussian_splatting_rasterizer.gd:185-193 for a camera at (0,0,5) looking down -Z."""
    vp, _ = default_vp(1920, 1080)
    v, p = vp[:16].reshape(4, 4), vp[16:].reshape(4, 4)   # rows here = columns of the matrix
    np.testing.assert_array_equal(v, np.float32([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, -5, 1]]))
    cot = 1 / np.tan(np.radians(37.5))
    np.testing.assert_allclose([p[0, 0], p[1, 1]], [cot / (1920 / 1080), cot], rtol=1e-6)
    assert p[2, 3] == -1 and p[3, 3] == 0 and p[0, 3] == 0 and p[1, 3] == 0
    np.testing.assert_allclose([p[2, 2], p[3, 2]], [-(4000.05) / 3999.95, -2 * 0.05 * 4000 / 3999.95], rtol=1e-6)


@pytest.mark.parametrize("w,h", [(17, 17), (16, 16), (1, 1), (33, 15)])
def test_odd_sizes_render(w, h):
    case = make_case(300, w, h, seed=9, scale_n=3000)
    out = oracle.render_frame(case["records"], oracle_frame(case))
    assert out["image"].shape == (h, w, 4) and np.all(out["image"][..., 3] == 1)


def test_tiny_isotropic_splats_fail_the_eigenvalue_test():
    """gsplat_projection.glsl:181-182: sqrt(max(0.1, mid^2 - det)) >= 0.316, so a near-isotropic footprint with
    mid < 0.316 (i.e. projected variance < ~0.016 px^2 on top of the 0.3 low-pass) gets a negative 'eigenvalue'
    and is dropped — faithful to the reference, surprising enough to pin."""
    vp, cam = default_vp(128, 96)
    small = record(cov=(1e-4, 0, 0, 1e-4, 0, 1e-4), opacity=0.9)[None]
    big = record(cov=(1e-3, 0, 0, 1e-3, 0, 1e-3), opacity=0.9)[None]
    fr = oracle.Frame.make(vp, cam, 128, 96)
    assert oracle.project(small, fr)["visible"] == 0
    assert oracle.project(big, fr)["visible"] == 1
