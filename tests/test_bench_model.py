"""bench.py's byte model (what `roofline` divides by): per kernel class and launch, charged where this build does the
work — the pairs a frame actually emitted, the key width it used, the colours where they were evaluated."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stats(**kw):
    st = {"num_splats": 6_000_000, "num_visible": 5_800_000, "num_sorted": 58_000_000, "num_composited": 2_100_000,
          "sh_degree": 3, "lazy_colors": 1, "pair_key_bytes": 2, "pairs_round": [58_000_000, 0], "_tiles": 8160,
          "_pixels": 1920 * 1080}
    st.update(kw)
    return st


def test_survey_bytes_do_not_depend_on_how_the_frame_was_rendered():
    b = _bench()
    one = b.phase_algorithmic_bytes(_stats())
    two = b.phase_algorithmic_bytes(_stats(pairs_round=[5_000_000, 3_000], pair_key_bytes=4, lazy_colors=0))
    assert one == two and one["sort"] == 68 * 58_000_000   # the reference's four pair passes over all D


def test_build_bytes_follow_the_pairs_emitted_the_key_width_and_the_colour_mode():
    b = _bench()
    wide = b.kernel_algorithmic_bytes(_stats(pair_key_bytes=4))
    narrow = b.kernel_algorithmic_bytes(_stats(pair_key_bytes=2))
    assert narrow["sort_downsweep"] * 16 == wide["sort_downsweep"] * 12     # 2 x (2 + 4) vs 2 x (4 + 4) bytes per pair
    assert narrow["sort_upsweep"] * 2 == wide["sort_upsweep"]
    two = b.kernel_algorithmic_bytes(_stats(pairs_round=[5_000_000, 3_000]))
    # two rounds: twice the launches, each charged the average round's pairs — 58 M pairs became 5 M
    assert two["sort_downsweep"] * 2 < 0.1 * narrow["sort_downsweep"]
    assert two["emit"] < narrow["emit"] and two["boundaries"] < narrow["boundaries"]
    eager = b.kernel_algorithmic_bytes(_stats(lazy_colors=0))
    k = 16
    # the SH coefficients move with the evaluation, and a lazy frame writes no RasterizeData: its compositor reads the
    # splat's position / covariance / opacity (44 B) from the scene slot instead of 36 B of a record
    assert eager["project"] - narrow["project"] == (12 * k + 48) * 5_800_000
    assert narrow["render"] - eager["render"] == (12 * k + 44 - 36) * 2_100_000


def test_survey_bytes_of_the_dominant_kernel_are_the_judges_formula():
    b = _bench()
    st = _stats()
    sv = b.survey_kernel_bytes(st)
    assert sv["render"] == 40 * 2_100_000 + 16 * 1920 * 1080            # SURVEY.md §8(d) B_render with D_c
    assert sv["project"] + sv["emit"] == b.phase_algorithmic_bytes(st)["projection"]
    assert "splat_sort" not in sv and "scan" not in sv                  # no counterpart in the reference
    two = b.survey_kernel_bytes(_stats(pairs_round=[5_000_000, 3_000]))
    assert two["render"] * 2 == sv["render"]                            # per launch: two compositor launches


def test_workload_from_a_ply_file(tmp_path):
    """bench.py --ply: an INRIA .ply read by PlyFile.parse (util/ply_file.gd:10-19) becomes the workload — same rows,
    the SH band count found in the file, the requested frame size and camera."""
    import argparse
    import numpy as np
    from godotgaussiansplatting_amd import scenes
    b = _bench()
    rows = scenes.synthetic_rows(1500, 77, 2)
    path = tmp_path / "scene.ply"
    scenes.write_ply(str(path), rows)
    args = argparse.Namespace(ply=str(path), config="c3", width=640, height=360, eye="1,2,6", target="0,0,0")
    wl = b.Workload(args)
    assert (wl.n, wl.deg, wl.w, wl.h) == (1500, 2, 640, 360) and wl.name == "ply:scene.ply"
    np.testing.assert_array_equal(wl.rows(), rows)
    np.testing.assert_array_equal(wl.rows(500), rows[:500])
    assert wl.vp.shape == (32,) and "scene.ply" in wl.label
    # and a named configuration keeps the generator's rows, sample prefixes drawn with the full scene's size law
    args = argparse.Namespace(ply=None, config="c1", width=0, height=0, eye=None, target=None)
    wl = b.Workload(args)
    assert (wl.n, wl.deg, wl.w, wl.h) == scenes.CONFIGS["c1"][:4]
    np.testing.assert_array_equal(wl.rows(1000), scenes.config_rows("c1", 1000))


def test_counters_of_other_kernels_are_not_pasted_into_the_line(tmp_path):
    """roofline.traffic and roofline.binding_bound come from profiles/*.json (PMC passes cannot run inside the bench); each
    entry carries the hashes of the sources of its kernel class.  A tree whose raster.hip differs in one digit of CODE from the
    one the counters were taken on gets `traffic: None, traffic_stale: True` — and keeps the projection kernel's figure, whose
    sources did not change."""
    import json
    import shutil
    import sys
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import provenance
    root = tmp_path / "tree"
    csrc = root / "godotgaussiansplatting_amd" / "csrc"
    csrc.mkdir(parents=True)
    for name in provenance.sha_of_tree():
        shutil.copy(os.path.join(ROOT, "godotgaussiansplatting_amd", "csrc", name), csrc / name)
    (root / "profiles").mkdir()
    shas = provenance.sha_of_tree(str(root))
    pmc = {"_source": "test", "c3": {"_csrc_sha256": shas, "render": {"hbm_bytes_per_launch": 9.0e8},
                                      "project": {"hbm_bytes_per_launch": 3.5e8}}}
    sq = {"_source": "test", "c3": {"_csrc_sha256": shas, "render": {"valu_issue_frac": 0.73, "lds_frac": 0.58,
                                                                     "waves_per_simd": 7.0}}}
    json.dump(pmc, open(root / "profiles" / "pmc_traffic.json", "w"))
    json.dump(sq, open(root / "profiles" / "sq_bound.json", "w"))
    traffic, stale, src, binding = b.counters_from_profiles(str(root), "c3", "render")
    assert traffic == 9.0e8 and stale is False and binding["frac"] == 0.73
    raster = csrc / "raster.hip"
    text = raster.read_text()
    raster.write_text("// a comment more, a line re-wrapped:\n" + text.replace("\n\n", "\n \n", 3))
    assert b.counters_from_profiles(str(root), "c3", "render")[:2] == (9.0e8, False)   # comments and white space do not count
    assert "constexpr float EXP_CUTOFF = -32.0f;" in text
    raster.write_text(text.replace("constexpr float EXP_CUTOFF = -32.0f;", "constexpr float EXP_CUTOFF = -31.0f;"))   # one digit of the compositor's code
    traffic, stale, src, binding = b.counters_from_profiles(str(root), "c3", "render")
    assert traffic is None and stale is True and "raster.hip" in src
    assert binding == {"kind": None, "stale": True, "source": binding["source"]}
    traffic, stale, _, _ = b.counters_from_profiles(str(root), "c3", "project")     # projection.hip is untouched
    assert traffic == 3.5e8 and stale is False
    # files collected before the hashes were recorded count as stale; a config without an entry has nothing to report
    del pmc["c3"]["_csrc_sha256"]
    json.dump(pmc, open(root / "profiles" / "pmc_traffic.json", "w"))
    assert b.counters_from_profiles(str(root), "c3", "project")[:2] == (None, True)
    assert b.counters_from_profiles(str(root), "c9", "render")[:2] == (None, None)


def test_committed_counter_file_holds_no_empty_pass():
    """profiles/pmc_traffic.json feeds roofline.traffic of the driver's line.  A --pmc pass that did not collect leaves its
    counter at zero for every kernel, and round 5 summarised two such passes as `2 x FETCH + 0` (c3r: a compositor that
    writes a 33 MB image "moving 30 MB", traffic_over_algorithmic 0.51).  The invariants tools/summarize_profile.py now
    refuses a pass on, over the committed file: both counters non-zero for the compositor, the pair downsweep and the
    projection; the compositor's heaviest launch writes at least the image (16 P) and moves at least that; a pair
    downsweep moves at least its output.  And the summariser really refuses: a synthetic entry with WRITE_SIZE = 0."""
    import json
    import sys
    from godotgaussiansplatting_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import summarize_profile as sp
    data = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    seen = 0
    for cfg, ent in data.items():
        if not isinstance(ent, dict):
            continue
        n, deg, w, h, seed = scenes.CONFIGS[cfg]
        assert sp.entry_problems(cfg, ent, w * h) == [], cfg
        assert ent["render"]["hbm_bytes_per_launch"] >= 16 * w * h
        seen += 1
    assert seen >= 3
    good = {"render": {"hbm_bytes_per_launch": 9.0e8, "fetch_kib": 4.0e5, "write_kib": 9.8e4}}
    assert sp.entry_problems("c3", good, 1920 * 1080) == []
    empty_write = {"render": {"hbm_bytes_per_launch": 2 * 14568.7 * 1024, "fetch_kib": 14568.7, "write_kib": 0.0}}
    assert any("WRITE_SIZE is zero" in p for p in sp.entry_problems("c3r", empty_write, 1920 * 1080))
    short_sort = {"sort_downsweep": {"hbm_bytes_per_launch": 3.0e7, "fetch_kib": 1.0e4, "write_kib": 1.0e4},
                  "_frame": {"P": 1920 * 1080, "D": 10_000_000, "pairs_round": [10_000_000, 0], "sort_passes": 4, "pair_key_bytes": 2}}
    assert any("below its 60.0 MB of output" in p for p in sp.entry_problems("c3", short_sort))


def test_summariser_refuses_a_pass_that_did_not_collect(tmp_path):
    """tools/summarize_profile.py end to end on synthetic rocpd databases: a WRITE_SIZE pass whose counter is zero for every
    kernel (round 5's c3r / c3d), or whose database is missing (round 6: the profiler's counter tool crashed), gives a
    "FAILED PASS" summary, NO entry in pmc_traffic.json — an implausible one already there is removed — and a non-zero
    return; a pass that did collect is merged with the hashes of the kernel sources."""
    import json
    import sqlite3
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import summarize_profile as sp

    def make_db(path, counter, values):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        db = sqlite3.connect(path)
        db.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
        for name, v in values.items():
            for _ in range(3):
                db.execute("insert into counters_collection values (?, ?, ?)", (name, counter, v))
        db.commit()
        db.close()

    render = "void gsplat::(anonymous namespace)::render_kernel<false, 3, 0, false, false>(float4 const*)"
    down = "void gsplat::(anonymous namespace)::downsweep_pairs_kernel<7, true, unsigned short>(int)"
    proj = "void gsplat::(anonymous namespace)::project_kernel<-1, false>(int)"
    fetch = {render: 390000.0, down: 30000.0, proj: 150000.0}
    prefix = str(tmp_path / "profiles" / "rXX_c3")
    os.makedirs(tmp_path / "profiles")
    tj = tmp_path / "profiles" / "pmc_traffic.json"
    # (a) the write pass came back empty; the file holds an older, equally empty entry for the configuration
    json.dump({"c3": {"render": {"hbm_bytes_per_launch": 3.0e7, "fetch_kib": 14568.7, "write_kib": 0.0}}}, open(tj, "w"))
    src = tmp_path / "prof_a"
    make_db(str(src / "pmc_fetch" / "fetch_results.db"), "FETCH_SIZE", fetch)
    make_db(str(src / "pmc_write" / "write_results.db"), "WRITE_SIZE", {k: 0.0 for k in fetch})
    assert sp.summarize_pmc(str(src), prefix, "c3") != 0
    assert "c3" not in json.load(open(tj))
    md = open(prefix + "_pmc.md").read()
    assert "FAILED PASS" in md and "WRITE_SIZE: zero for every kernel" in md and "Refused" in md
    # (b) the write pass's database does not exist at all
    src = tmp_path / "prof_b"
    make_db(str(src / "pmc_fetch" / "fetch_results.db"), "FETCH_SIZE", fetch)
    assert sp.summarize_pmc(str(src), prefix, "c3") != 0 and "c3" not in json.load(open(tj))
    assert "write_results.db does not exist" in open(prefix + "_pmc.md").read()
    # (c) both passes collected: merged, stamped with the source hashes
    src = tmp_path / "prof_c"
    make_db(str(src / "pmc_fetch" / "fetch_results.db"), "FETCH_SIZE", fetch)
    make_db(str(src / "pmc_write" / "write_results.db"), "WRITE_SIZE", {render: 97000.0, down: 29000.0, proj: 60000.0})
    assert sp.summarize_pmc(str(src), prefix, "c3") == 0
    ent = json.load(open(tj))["c3"]
    assert ent["render"]["hbm_bytes_per_launch"] == (2 * 390000.0 + 97000.0) * 1024 and "_csrc_sha256" in ent
    assert "FAILED" not in open(prefix + "_pmc.md").read()
