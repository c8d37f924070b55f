"""The HIP path against an INDEPENDENT witness: oracle/numpy_twin.py, a float64 evaluation of the literal GLSL that
shares nothing with the C oracle the other GPU tests compare with (no arithmetic contract, no polynomial exp, no
re-association) — so agreement here does not rest on two co-designed twins agreeing with each other.

* gsplat_projection.glsl:150-226: which splats survive, their tile rectangles and 16-bit depth codes — bit-exact
  wherever the twin's own perturbation analysis says a binary32 evaluation has no choice (tests/twin_checks.py);
* radix_sort_*.glsl: the sorted pairs of the frame equal the result of the literal, invocation-by-invocation emulation
  of the three sort shaders at subgroup size 32 (oracle/radix_glsl.py) run on the projection's emission;
* gsplat_boundaries.glsl:23-50: tile ranges equal the thread-by-thread evaluation on the frame's own sorted keys;
* gsplat_projection.glsl:202-206: RasterizeData to binary32 rounding of the twin's float64 records;
* gsplat_render.glsl:50-101: RGBA within the north-star 1e-4 of the twin's compositor reading the frame's own
  RasterizeData (a binary32 buffer in the reference too) and tile lists, except knife-edge pixels — and which pixels
  are knife edges is decided by the twin alone; the end-to-end error (twin's own float64 records) is reported.
The reports (fractions of unstable decisions, knife-edge pixels, worst errors) are printed and, on the GPU box, written
to gpurun_out/twin_report_<name>.json.
"""
import json
import os

import numpy as np
import pytest

from conftest import hip_frame, make_case

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, rep):
    print(name, json.dumps(rep))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"twin_report_{name}.json"), "w") as f:
            json.dump(rep, f, indent=1)


def test_config1_size_against_the_literal_glsl_twin():
    """BASELINE.json configs[0] geometry (1280x720, 100 k splats; demo.ply itself is not available): whole frame."""
    import oracle
    import twin_checks as tc
    from oracle import numpy_twin as twin
    from oracle import radix_glsl as rg
    from godotgaussiansplatting_amd import capi, scenes
    n, deg, w, h, seed = scenes.CONFIGS["c1"]
    case = make_case(n, w, h, seed=seed, sh_degree=deg)
    gx, gy = (w + 15) // 16, (h + 15) // 16
    with capi.Context(n, w, h) as ctx:
        ctx.upload_splats(case["records"])
        img = ctx.render_to_host(hip_frame(case))
        counts = ctx.read_counts()
        culled = ctx.read_culled()
        sk, sv = ctx.read_sorted()
        bounds = ctx.read_bounds()
    p = tc.project_chunked(case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h)
    rep = tc.check_integer_decisions(p, counts, sk, sv, n)
    assert rep["compared_rects"] > 0.9 * rep["visible"] and rep["unstable_cull_or_rect_frac"] < 5e-3
    # the frame's pairs in the reference's emission order (ascending splat id, y outer / x inner,
    # gsplat_projection.glsl:219-226) through the literal sort shaders
    order = np.lexsort((np.arange(sk.size), sv))          # by splat id; a splat's tiles ascend in (y, x) already
    lk, lv = rg.sort_pairs(sk[order], sv[order])
    np.testing.assert_array_equal(lk, sk)
    np.testing.assert_array_equal(lv, sv)
    np.testing.assert_array_equal(bounds, twin.boundaries(sk, gx * gy))
    full = (0, gx, 0, gy)
    ids = tc.splats_in_tiles(sv, bounds, gx, full)
    rep["records"] = tc.check_records(culled, case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h,
                                      case["time"], ids)
    rep["image"] = tc.check_image(culled, w, h, 0.0, img, sv, bounds, full)
    own = tc.twin_records(case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h, case["time"], ids)
    rep["image_end_to_end"] = tc.check_image(own, w, h, 0.0, img, sv, bounds, full, tol=5e-3)
    rep["pairs"] = int(sk.size)
    _report("c1", rep)


@pytest.mark.parametrize("kw", [dict(model_scale=1.3, time=0.6, load_time=0.0, heatmap=1.0), dict(sh_degree=3)],
                         ids=["animation-heatmap", "deg3"])
def test_small_frames_against_the_literal_glsl_twin(kw):
    """Load animation + model scale + heat map, and full SH, at a size where every tile is looked at."""
    import twin_checks as tc
    from godotgaussiansplatting_amd import capi
    n, w, h = 30000, 640, 360
    case = make_case(n, w, h, seed=501, scale_n=4000, **kw)
    gx, gy = (w + 15) // 16, (h + 15) // 16
    with capi.Context(n, w, h, key_budget_factor=100) as ctx:
        ctx.upload_splats(case["records"])
        for _ in range(3):
            img = ctx.render_to_host(hip_frame(case))
        assert ctx.stats()["overflow"] == 0
        counts = ctx.read_counts()
        culled = ctx.read_culled()
        sk, sv = ctx.read_sorted()
        bounds = ctx.read_bounds()
    p = tc.project_chunked(case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h, time=case["time"])
    rep = tc.check_integer_decisions(p, counts, sk, sv, n)
    full = (0, gx, 0, gy)
    ids = tc.splats_in_tiles(sv, bounds, gx, full)
    rep["records"] = tc.check_records(culled, case["records"], case["vp"], case["cam_pos"], case["model_scale"], w, h,
                                      case["time"], ids)
    rep["image"] = tc.check_image(culled, w, h, case["heatmap"], img, sv, bounds, full)
    _report("small_" + "_".join(sorted(kw)), rep)
