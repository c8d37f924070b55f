"""Host-side mirror of the reference's GDScript interface (PlyFile, GaussianSplattingRasterizer), the synthetic
scene generator and the stripe partition logic.  CPU only — nothing here launches a kernel."""
import numpy as np
import pytest

from godotgaussiansplatting_amd import Camera3D, GaussianSplattingRasterizer, PlyFile, Texture2DRD, scenes
from godotgaussiansplatting_amd.distributed import StripeLayout, balanced_cuts, even_cuts, unstripe


def test_synthetic_rows_are_deterministic_and_follow_the_spec():
    a = scenes.synthetic_rows(5000, 42, 3)
    b = scenes.synthetic_rows(5000, 42, 3)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (5000, 62) and a.dtype == np.float32
    assert np.all(a[:, 3:6] == 0)                                   # normals unused
    s_lo = 0.002 * (1e6 / 5000) ** (1 / 3)
    assert np.all(a[:, 55:58] >= np.log(s_lo) - 1e-5) and np.all(a[:, 55:58] <= np.log(10 * s_lo) + 1e-5)
    assert abs(a[:, 0:3].std() - 1.5) < 0.05 and abs(a[:, 54].std() - 2.0) < 0.1
    assert np.all(np.abs(a[:, 6:9]) <= 1.5)
    d0 = scenes.synthetic_rows(5000, 42, 0)
    assert np.all(d0[:, 9:54] == 0)
    np.testing.assert_array_equal(d0[:, :9], a[:, :9])              # same draw order up to f_dc
    assert scenes.CONFIGS["c2"] == (1_000_000, 0, 1920, 1080, 2)


def test_ply_roundtrip(tmp_path):
    rows = scenes.synthetic_rows(1234, 7, 3)
    path = str(tmp_path / "scene.ply")
    scenes.write_ply(path, rows)
    ply = PlyFile(path)
    assert ply.size == 1234 and len(ply.properties) == 62
    assert ply.properties[:3] == ["x", "y", "z"] and ply.properties[54] == "opacity" and ply.properties[58] == "rot_0"
    np.testing.assert_array_equal(ply.rows(), rows)
    v = ply.get_vertex(10)
    assert v["x"] == rows[10, 0] and v["scale_2"] == rows[10, 57] and v["rot_3"] == rows[10, 61]
    mem = PlyFile.from_rows(rows)
    assert mem.size == 1234 and mem.properties == ply.properties


def test_ply_big_endian_and_errors(tmp_path):
    rows = scenes.synthetic_rows(10, 1, 0)
    names = PlyFile.from_rows(rows).properties
    header = "ply\nformat binary_big_endian 1.0\nelement vertex 10\n" + "".join(f"property float {n}\n" for n in names)
    p = tmp_path / "be.ply"
    p.write_bytes((header + "end_header\n").encode() + rows.astype(">f4").tobytes())
    np.testing.assert_array_equal(PlyFile(str(p)).rows(), rows)
    q = tmp_path / "short.ply"
    q.write_bytes((header + "end_header\n").encode() + rows.astype(">f4").tobytes()[:100])
    with pytest.raises(ValueError):
        PlyFile(str(q))


def test_loader_chunks_cover_the_scene_and_report_progress():
    rows = scenes.synthetic_rows(2500, 3, 0)
    ply = PlyFile.from_rows(rows)
    got = np.zeros_like(rows)
    calls = []

    class Dev:
        def buffer_update_ply_rows(self, buffer, first, chunk, creation_time):
            got[first:first + chunk.shape[0]] = chunk
            calls.append((first, chunk.shape[0], creation_time))

    done, loaded = [], [0]
    PlyFile.load_gaussian_splats(ply, ply.size // 1000, Dev(), None, [False], loaded, lambda: done.append(1),
                                 time_source=lambda: 12.5)
    np.testing.assert_array_equal(got, rows)
    assert loaded[0] == 2500 and done == [1]
    assert all(c[1] <= 2 for c in calls) and all(c[2] == 12.5 for c in calls)   # stride = size/1000
    # scenes under 1000 splats: the reference divides by zero; here stride clamps to 1
    small = PlyFile.from_rows(rows[:10])
    loaded = [0]
    PlyFile.load_gaussian_splats(small, small.size // 1000, Dev(), None, [False], loaded, lambda: None)
    assert loaded[0] == 10
    # cancellation flag (ply_file.gd:35,70)
    loaded = [0]
    PlyFile.load_gaussian_splats(ply, 100, Dev(), None, [True], loaded, lambda: None)
    assert loaded[0] == 0


def test_rasterizer_mirror_surface_without_gpu():
    rows = scenes.synthetic_rows(2000, 1, 0)
    cam = Camera3D.from_spec(scenes.default_camera(), 1280 / 720)
    tex = Texture2DRD()
    r = GaussianSplattingRasterizer(PlyFile.from_rows(rows), (1280, 720), tex, cam)
    # gaussian_splatting_rasterizer.gd:26-29
    assert r.texture_size == (1280, 720) and r.tile_dims == (80, 45)
    r.render_scale[0] = 0.5
    r.texture_size = (1281, 721)
    assert r.texture_size == (640, 360) and r.tile_dims == (40, 23)
    r.texture_size = (1, 1)
    assert r.texture_size == (1, 1) and r.tile_dims == (1, 1)
    for name in ("init_gpu", "cleanup_gpu", "rasterize", "get_splat_position", "update_camera_matrices", "loaded",
                 "render_scale", "model_scale", "should_enable_heatmap", "basis_override", "is_loaded",
                 "num_splats_loaded", "should_terminate_thread", "point_cloud", "camera", "render_texture"):
        assert hasattr(r, name), name
    assert (r.TILE_SIZE, r.WORKGROUP_SIZE, r.RADIX, r.PARTITION_SIZE) == (16, 512, 256, 4096)
    # update_camera_matrices: True when the camera moved, False otherwise (:175-195)
    assert r.update_camera_matrices() is True
    assert r.update_camera_matrices() is False
    v = r.camera_push_constants[:16].reshape(4, 4)
    np.testing.assert_array_equal(v[3], np.float32([0, 0, -5, 1]))
    cam.global_position = np.float32([1, 2, 7])
    assert r.update_camera_matrices() is True
    np.testing.assert_array_equal(r.camera_push_constants[12:16], np.float32([-1, 2, -7, 1]))  # (-O.X, O.Y, -O.Z, 1)
    np.testing.assert_array_equal(r._uniform_cam_pos, np.float32([-1, -2, 7]))
    with pytest.raises(RuntimeError):
        r.get_splat_position((10, 10))
    r.cleanup_gpu()
    assert tex.texture_rd_rid == 0


def test_stripe_cuts_and_layout():
    assert even_cuts(120, 8) == [0, 15, 30, 45, 60, 75, 90, 105, 120]
    assert even_cuts(5, 8)[0] == 0 and even_cuts(5, 8)[-1] == 5
    w = np.array([1, 1, 1, 1, 10, 10, 1, 1, 1, 1], float)
    cuts = balanced_cuts(w, 2)
    assert cuts == [0, 5, 10]
    cuts = balanced_cuts(np.exp(-0.5 * ((np.arange(120) - 60) / 15.0) ** 2) + 0.05, 8)
    assert cuts[0] == 0 and cuts[-1] == 120 and all(b > a for a, b in zip(cuts, cuts[1:]))
    widths = np.diff(cuts)
    assert widths[3] < widths[0] and widths[4] < widths[7]           # narrow stripes where the work is
    lay = StripeLayout("columns", 1920, 1080, even_cuts(120, 8))
    assert lay.slot_shape() == (1080, 240, 4) and lay.slot_origin(3) == (720, 0)
    lay = StripeLayout("columns", 1000, 100, [0, 20, 63])             # ragged: 320 px and 680 px (clipped at 1000)
    assert lay.px_range(1) == (320, 1000) and lay.slot_px == 680
    lay_r = StripeLayout("rows", 640, 360, [0, 10, 23])
    assert lay_r.slot_shape() == (200, 640, 4) and lay_r.px_range(1) == (160, 360)


@pytest.mark.parametrize("axis", ["columns", "rows"])
def test_unstripe_reassembles_the_frame(axis):
    rng = np.random.default_rng(0)
    w, h = 200, 120
    full = rng.random((h, w, 4)).astype(np.float32)
    n = (w + 15) // 16 if axis == "columns" else (h + 15) // 16
    lay = StripeLayout(axis, w, h, [0, 2, 3, n])
    staging = np.zeros((3,) + lay.slot_shape(), np.float32)
    for r in range(3):
        a, b = lay.px_range(r)
        if axis == "columns":
            staging[r, :, : b - a] = full[:, a:b]
        else:
            staging[r, : b - a] = full[a:b]
    out = unstripe(staging, lay, np.zeros_like(full))
    np.testing.assert_array_equal(out, full)


def test_time_balanced_cuts_converge_on_equal_rank_times():
    """bench.py --gpus N re-cuts the stripes from the ranks' MEASURED frame times (distributed.time_balanced_cuts).  A rank's
    time here: a fixed part, the pairs of its tile rows, and a part that grows with the stripe's WIDTH (the splats a wide,
    sparse edge stripe has to project) — the term a pairs-only balance cannot see.  From the pairs-balanced cuts a few
    iterations bring max / mean of the rank times under 1.05; equal times are a fixed point."""
    import numpy as np
    from godotgaussiansplatting_amd.distributed import balanced_cuts, time_balanced_cuts
    rows = 135
    x = (np.arange(rows) - rows / 2) / (rows / 6)
    pairs = 2.0e5 * np.exp(-0.5 * x * x) + 500.0        # a dense centre, sparse edges (c4's profile)

    def rank_times(cuts):
        t = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            t.append(0.12 + 6.0e-8 * pairs[a:b].sum() + 4.0e-3 * (b - a))   # ms: fixed + pairs + width (projection)
        return np.array(t)

    world = 8
    cuts = balanced_cuts(pairs + 64.0 * 240, world)
    first = rank_times(cuts)
    assert first.max() / first.mean() > 1.15            # pairs-balanced: the wide edge stripes are slow
    for _ in range(4):
        cuts = time_balanced_cuts(cuts, rank_times(cuts), prior=pairs + 64.0 * 240)
        assert cuts[0] == 0 and cuts[-1] == rows and all(b > a for a, b in zip(cuts[:-1], cuts[1:]))
    last = rank_times(cuts)
    assert last.max() / last.mean() < 1.05 and last.max() < 0.9 * first.max()
    assert time_balanced_cuts(cuts, np.full(world, 0.3), prior=None) is not None
    # equal times with a flat prior keep the cuts where they are
    even = [0, 10, 20, 30, 40]
    assert time_balanced_cuts(even, [0.2, 0.2, 0.2, 0.2]) == even
    # a rank without tiles (cuts may repeat) is tolerated
    assert len(time_balanced_cuts([0, 5, 5, 12], [0.3, 0.0, 0.3])) == 4


def test_batch_submitter_renders_exactly_the_frames_it_was_given():
    """bench.py --gpus N --batch B: frames are handed over one by one and go out B at a time, round-robin over the launch
    sequences in flight; what is left when the run ends goes out as a partial batch — K steps are K frames."""
    from godotgaussiansplatting_amd.distributed import BatchSubmitter

    class FakeGroup:
        def __init__(self):
            self.calls = []

        def render(self, frame):
            self.calls.append(("render", [frame]))

        def render_batch(self, frames):
            self.calls.append(("render_batch", list(frames)))

    for batch, in_flight, k in ((4, 3, 20), (4, 3, 22), (3, 2, 7), (1, 3, 5), (4, 1, 3)):
        groups = [FakeGroup() for _ in range(in_flight)]
        sub = BatchSubmitter(groups, batch)
        for i in range(k):
            sub.step(i)
        sub.flush()
        sub.flush()                                            # (nothing left: no empty submission)
        assert sub.frames_submitted == k and sub.submissions == -(-k // batch)
        order = []
        turn = 0
        while any(g.calls for g in groups):                    # round-robin: submission j went to group j % in_flight
            kind, frames = groups[turn % in_flight].calls.pop(0)
            assert kind == ("render_batch" if batch > 1 else "render") and 1 <= len(frames) <= batch
            order += frames
            turn += 1
        assert order == list(range(k))                         # every frame once, in order
