"""The C oracle against an independent float64 NumPy evaluation of the literal GLSL (oracle/numpy_twin.py).

Neither is the reference (which cannot run here), but they were written separately — one from the arithmetic
contract, one from the shader text in ideal arithmetic — so agreement pins both.  CPU only.
"""
import numpy as np
import pytest

import oracle
from oracle import numpy_twin as twin
from conftest import make_case, oracle_frame


@pytest.mark.parametrize("n,w,h,seed,deg,kw", [
    (3000, 160, 96, 3, 3, {}),
    (2500, 131, 77, 4, 1, {"model_scale": 1.3}),
    (2000, 128, 96, 5, 0, {"time": 0.7, "load_time": 0.0}),
])
def test_projection_matches_literal_glsl(n, w, h, seed, deg, kw):
    case = make_case(n, w, h, seed=seed, sh_degree=deg, scale_n=20000, **kw)
    ref = oracle.render_frame(case["records"], oracle_frame(case))
    p = twin.project(case["records"], case["vp"][:16], case["vp"][16:], case["cam_pos"], case["model_scale"], w, h,
                     time=case["time"])
    alive_c = ref["counts"] > 0
    # survivors agree except splats sitting on a discontinuity
    assert np.mean(alive_c != p["alive"]) < 2e-3
    both = alive_c & p["alive"]
    assert both.sum() > 0.5 * n
    same_count = ref["counts"][both] == p["count"][both]
    assert same_count.mean() > 0.995
    np.testing.assert_allclose(ref["culled"][both], p["raster"][both], rtol=3e-5, atol=3e-5)
    # keys: depth code and tiles of splats whose rectangle agrees
    keys_u, vals_u, keys_s, vals_s = twin.emit_and_sort(p)
    agree = both.copy()
    agree[both] = same_count
    ids = np.nonzero(agree)[0]
    sel_c = np.isin(ref["values_unsorted"], ids)
    sel_t = np.isin(vals_u, ids)
    np.testing.assert_array_equal(ref["values_unsorted"][sel_c], vals_u[sel_t])
    kc, kt = ref["keys_unsorted"][sel_c], keys_u[sel_t]
    np.testing.assert_array_equal(kc >> 16, kt >> 16)                       # tile ids bit-exact
    dd = np.abs((kc & 0xFFFF).astype(np.int64) - (kt & 0xFFFF).astype(np.int64))
    assert dd.max() <= 1 and (dd != 0).mean() < 0.02                        # depth code: float32 vs float64 floor


def test_boundaries_match_thread_by_thread_evaluation():
    case = make_case(4000, 200, 120, seed=6, scale_n=20000)
    ref = oracle.render_frame(case["records"], oracle_frame(case))
    T = ref["bounds"].shape[0]
    np.testing.assert_array_equal(ref["bounds"], twin.boundaries(ref["keys"], T))
    # a frame whose last tile is populated exercises the D-1 rule
    k = np.sort(np.concatenate([ref["keys"], np.uint32([(T - 1) << 16] * 3)]))
    np.testing.assert_array_equal(oracle.boundaries(k, T), twin.boundaries(k, T))


@pytest.mark.parametrize("n,w,h,seed,scale_n,heat", [(2500, 96, 64, 7, 4000, 0.0), (2500, 96, 64, 8, 4000, 1.0),
                                                     (6000, 144, 80, 9, 600, 0.0), (1500, 80, 48, 10, 60, 0.0),
                                                     (20000, 320, 180, 11, 2500, 0.0)])
def test_compositor_matches_literal_glsl(n, w, h, seed, scale_n, heat):
    """Same RasterizeData, sorted values and tile ranges into both compositors: the contract's reassociated
    quadratic form, polynomial exp and t - alpha*t must stay within float32 noise of the literal float64
    expressions, except pixels that sit on the t <= 1/255 / block-sum discontinuities.  Which pixels those are is
    decided by the TWIN alone (its own result under a +-4e-6 perturbation of every alpha) — and, separately, by the
    oracle's own perturbation; both masks are small and the frames agree outside either.  The last two scenes have
    tiles of several batches that leave early (the block-sum rule decides what is composited at all)."""
    case = make_case(n, w, h, seed=seed, scale_n=scale_n, heatmap=heat)
    fr = oracle_frame(case)
    ref = oracle.render_frame(case["records"], fr, capacity=400 * n)
    assert ref["stats"]["overflow"] == 0
    raster = ref["culled"].astype(np.float64)
    img_t = twin.render(raster, ref["values"], ref["bounds"], w, h, heatmap_factor=heat)
    lo_t = twin.render(raster, ref["values"], ref["bounds"], w, h, heatmap_factor=heat, alpha_scale=1 - 4e-6)
    hi_t = twin.render(raster, ref["values"], ref["bounds"], w, h, heatmap_factor=heat, alpha_scale=1 + 4e-6)
    knife_twin = np.max(np.abs(hi_t - lo_t), axis=-1) > 2e-5
    lo, _, _ = oracle.render_tiles(ref["culled"], ref["values"], ref["bounds"], fr, exp_scale=1 - 4e-6)
    hi, _, _ = oracle.render_tiles(ref["culled"], ref["values"], ref["bounds"], fr, exp_scale=1 + 4e-6)
    knife_oracle = np.max(np.abs(hi - lo), axis=-1) > 2e-5
    assert knife_twin.mean() < 0.02 and knife_oracle.mean() < 0.02
    err = np.max(np.abs(ref["image"] - img_t), axis=-1)
    assert err[~knife_twin].max() < 2e-5      # the independent mask alone is enough
    assert err[~knife_oracle].max() < 2e-5
    assert np.all(ref["image"][..., 3] == 1.0)
    if scale_n <= 600:
        b = ref["bounds"].astype(np.int64)
        assert (np.clip(b[:, 1] - b[:, 0], 0, None) > 256).sum() >= 4 and ref["stats"]["composited"] < ref["D"]


@pytest.mark.parametrize("n,w,h,seed,deg,scale_n", [(20000, 320, 180, 21, 3, 2500), (8000, 200, 120, 22, 1, 20000)])
def test_frame_against_the_twin_end_to_end(n, w, h, seed, deg, scale_n):
    """The check the -m gpu twin tests run on the HIP path (tests/twin_checks.py), here with the C oracle as the
    producer: the twin projects for itself, the producer's rectangles / depth codes must be the twin's wherever the
    twin calls them stable, and the twin's compositor on its OWN float64 records reproduces the image."""
    import twin_checks as tc
    case = make_case(n, w, h, seed=seed, sh_degree=deg, scale_n=scale_n)
    ref = oracle.render_frame(case["records"], oracle_frame(case), capacity=400 * n)
    assert ref["stats"]["overflow"] == 0
    p = tc.project_chunked(case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h, chunk=7000)
    rep = tc.check_integer_decisions(p, ref["counts"], ref["keys"], ref["values"], n)
    assert rep["compared_rects"] > 0.5 * rep["visible"] and rep["unstable_cull_or_rect_frac"] < 5e-3
    gx, gy = oracle.grid(w, h)
    ids = tc.splats_in_tiles(ref["values"], ref["bounds"], gx, (0, gx, 0, gy))
    tc.check_records(ref["culled"], case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h, case["time"], ids)
    rep_img = tc.check_image(ref["culled"], w, h, 0.0, ref["image"], ref["values"], ref["bounds"], (0, gx, 0, gy))
    assert rep_img["max_err_off_knife_edges"] < 5e-5
    own = tc.twin_records(case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h, case["time"], ids)
    tc.check_image(own, w, h, 0.0, ref["image"], ref["values"], ref["bounds"], (0, gx, 0, gy), tol=5e-3)  # end to end
    np.testing.assert_array_equal(ref["bounds"], twin.boundaries(ref["keys"], gx * gy))
    # the same frame from the twin's OWN tile lists and depth order (what the GPU suite runs on a 1 % sample of the workload-
    # size frames): where the twin's list of a tile IS the producer's, only knife-edge pixels may differ by more than 1e-4
    own[ids, 0:2] = np.asarray(ref["culled"][ids, 0:2], np.float64)     # the frame's binary32 centres (see tests/test_gpu_full_size.py)
    oo = tc.own_order_sample(p, own, ref["image"], ref["values"], ref["bounds"], w, h, frac=0.25)
    assert oo["tiles_sampled"] >= 0.25 * oo["of_populated_tiles"] - 1
    ident = oo["per_class"]["identical_list"]
    assert ident["tiles"] > 0.5 * oo["tiles_sampled"] and ident["pixels_over_tol"] == ident["of_them_on_knife_edges"], oo
    assert ident["max_err_off_knife_edges"] <= 1e-4
