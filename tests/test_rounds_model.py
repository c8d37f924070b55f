"""The two-round frame schedule of the HIP build, as an executable NumPy model on the oracle's
projection: whatever the split, the pixels and the staged counts are the one-round frame's.  (CPU only; the GPU tests
check the same claim on the real kernels, against the oracle's image.)"""
import numpy as np
import pytest

from conftest import make_case, oracle_frame


def _proj(n, w, h, seed, scale_n, keep_top=None, distance=5.0):
    import oracle
    from godotgaussiansplatting_amd import scenes
    case = make_case(n, w, h, seed=seed, sh_degree=0, scale_n=scale_n, camera=scenes.default_camera(distance))
    if keep_top is not None:   # empty the lower part of the frame: the highest populated tile is not T - 1 (quirk Q5)
        full = oracle.project(case["records"], oracle_frame(case), capacity=400 * n)
        low = np.unique(full["values"][(full["keys"] >> 16) // ((w + 15) // 16) >= keep_top])
        case["records"][low, 0:3] = np.float32(1e6)
    p = oracle.project(case["records"], oracle_frame(case), capacity=400 * n)
    assert p["emitted"] == p["D"]
    return case, p


@pytest.mark.parametrize("scene", ["dense", "sparse", "top-only"])
def test_two_round_schedule_reproduces_the_one_round_frame(scene):
    from oracle import rounds_model as rm
    w, h = 96, 64
    gx, gy = 6, 4
    if scene == "dense":
        case, p = _proj(4000, w, h, 301, 10, distance=2.0)   # every tile holds several batches and saturates, T - 1 too
    elif scene == "sparse":
        case, p = _proj(1500, w, h, 302, 3000)      # few pairs per tile: nothing saturates
    else:
        case, p = _proj(1500, w, h, 303, 60, keep_top=2)
    last = int(p["keys"].max() >> 16) if p["D"] else 0
    assert (last == gx * gy - 1) == (scene != "top-only") and p["frame_last_tile_plus1"] == last + 1
    ref_img, ref_staged, d = rm.one_round(p, gx, gy, w, h)
    assert d == p["D"]
    if scene == "dense":
        assert (ref_staged < np.bincount(p["keys"] >> 16, minlength=gx * gy)).all()   # every tile exits early
    saved = False
    for frac in (0.0, 0.02, 0.1, 0.3, 0.6, 0.95, 1.0):
        for fix in (False, True):
            if fix:
                want_img, want_staged, _ = rm.one_round(p, gx, gy, w, h, fix_last=True)
            else:
                want_img, want_staged = ref_img, ref_staged
            img, staged, pairs = rm.two_rounds(p, gx, gy, w, h, frac, fix_last=fix)
            np.testing.assert_array_equal(img, want_img, err_msg=f"{scene} f={frac} fix={fix}")
            np.testing.assert_array_equal(staged, want_staged, err_msg=f"{scene} f={frac} fix={fix}")
            saved |= pairs < 0.7 * d
    assert saved == (scene == "dense")   # only where tiles saturate are pairs never emitted


def test_two_round_schedule_with_partial_edge_tiles():
    """W, H not multiples of 16 (1080 = 67.5 tiles): the out-of-image lanes of an edge tile are part of the early-exit
    sum, so their transmittance must survive the hand-over between the rounds.  With it, any split reproduces the
    one-round frame; without it (the round-2 defect, kept in the model as drop_offimage) an edge tile resumed by round B
    leaves its loop later and stages more pairs — which this test must be able to see."""
    from oracle import rounds_model as rm
    w, h = 88, 56                       # 5.5 x 3.5 tiles
    gx, gy = 6, 4
    case, p = _proj(4000, w, h, 311, 10, distance=2.0)
    ref_img, ref_staged, d = rm.one_round(p, gx, gy, w, h)
    edge = np.array([t for t in range(gx * gy) if t % gx == gx - 1 or t // gx == gy - 1])
    assert (ref_staged[edge] < np.bincount(p["keys"] >> 16, minlength=gx * gy)[edge]).any()   # edge tiles do exit early
    defect_seen = False
    for frac in (0.02, 0.05, 0.1, 0.2, 0.3, 0.6):
        img, staged, _ = rm.two_rounds(p, gx, gy, w, h, frac)
        np.testing.assert_array_equal(img, ref_img, err_msg=f"f={frac}")
        np.testing.assert_array_equal(staged, ref_staged, err_msg=f"f={frac}")
        _, staged_bad, _ = rm.two_rounds(p, gx, gy, w, h, frac, drop_offimage=True)
        inner = np.setdiff1d(np.arange(gx * gy), edge)
        np.testing.assert_array_equal(staged_bad[inner], ref_staged[inner])   # only edge tiles can tell
        assert (staged_bad >= ref_staged).all()
        defect_seen |= bool((staged_bad[edge] > ref_staged[edge]).any())
    assert defect_seen


def test_model_composites_what_the_oracle_composites():
    """The model's blend is not the arithmetic contract, but it is the same picture: its one-round frame stays within
    a few 1e-3 of the oracle's (so the schedule test above is about a real frame, not about zeros)."""
    import oracle
    from oracle import rounds_model as rm
    case, p = _proj(4000, 96, 64, 301, 10, distance=2.0)
    img, _, _ = rm.one_round(p, 6, 4, 96, 64)
    ref = oracle.render_frame(case["records"], oracle_frame(case), capacity=400 * case["records"].shape[0])
    assert ref["stats"]["overflow"] == 0 and float(np.abs(ref["image"]).max()) > 0.2
    assert float(np.abs(img - ref["image"]).max()) < 2e-2


def test_split_sort_on_tile_ids_is_the_reference_sort():
    """The sort this build runs instead of the reference's four 8-bit pair passes: order the visible
    SPLATS by (depth16, id), emit their pairs in that order, then sort the PAIRS by the 16-bit tile id alone, stably.
    Same array as a stable sort of the id-order emission on the whole 32-bit key — the oracle's."""
    import oracle
    case, _ = _proj(3000, 320, 192, 305, 300)
    ref = oracle.render_frame(case["records"], oracle_frame(case), capacity=400 * case["records"].shape[0])
    assert ref["stats"]["overflow"] == 0 and ref["D"] > 10000
    ku, vu = ref["keys_unsorted"], ref["values_unsorted"]           # the reference's emission: ascending splat id
    by_depth = np.argsort(ku & 0xFFFF, kind="stable")               # = by (depth16, id): this build's emission order
    tiles16 = (ku[by_depth] >> 16).astype(np.uint16)                # what the pair passes carry: 2 + 4 bytes per pair
    by_tile = np.argsort(tiles16, kind="stable")
    np.testing.assert_array_equal(vu[by_depth][by_tile], ref["values"])
    np.testing.assert_array_equal(ku[by_depth][by_tile], ref["keys"])   # (the taps rebuild the key from tile and depth16)
