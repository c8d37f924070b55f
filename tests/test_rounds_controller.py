"""gsplat::RoundsController (csrc/rounds_controller.h) against a simulated GPU: does a context end up on the right frame
schedule, how fast, and what do its trials cost?  (The controller is HIP-free; api.hip feeds it real frame times.)"""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ctl") / "rounds_controller_sim")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I",
                           os.path.join(ROOT, "godotgaussiansplatting_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "rounds_controller_sim.cpp"), "-o", exe])

    def run(scene, lag, frames, noise=0.004):
        return json.loads(subprocess.check_output([exe, scene, str(lag), str(frames), str(noise)]).decode())
    return run


@pytest.mark.parametrize("lag", [1, 2, 3])
def test_a_dense_scene_ends_up_on_two_rounds_near_the_best_fraction(sim, lag):
    r = sim("dense", lag, 4000)
    assert r["final_two"] == 1 and 0.008 <= r["final_frac"] <= 0.06, r
    assert 0 < r["first_hold_two"] <= 200, r                      # settled within 200 frames of the session's start
    assert r["tail_two_share"] >= 0.97, r                          # (the re-checks give one round a few frames)
    assert r["tail_ms"] <= 1.03 * r["tail_best_ms"], r             # within 3 % of the best schedule, trials included


@pytest.mark.parametrize("lag", [1, 2, 3])
def test_a_sparse_scene_stays_on_one_round_and_the_trials_stay_cheap(sim, lag):
    r = sim("sparse", lag, 4000)
    assert r["final_two"] == 0, r
    assert r["tail_two_share"] <= 0.03, r
    assert r["tail_ms"] <= 1.005 * r["tail_best_ms"], r


def test_a_scene_that_changes_is_noticed_by_the_next_recheck(sim):
    r = sim("dense_then_sparse", 1, 6000)
    assert r["final_two"] == 0 and r["tail_two_share"] <= 0.05, r


@pytest.mark.parametrize("scene", ["dense", "sparse"])
def test_a_host_far_ahead_of_the_gpu_cannot_learn_but_does_not_pay_either(sim, scene):
    """An unpaced loop (the host hundreds of frames ahead): frame times arrive after their trial is over and are
    ignored, the session stays on what it holds — one round — and the trials cost eight frames per candidate."""
    r = sim(scene, 300, 4000)
    assert r["final_two"] == 0 and r["tail_two_share"] <= 0.05, r


@pytest.mark.parametrize("scene,two", [("dense", 1), ("sparse", 0)])
def test_noisy_frame_times_do_not_flip_a_clear_decision(sim, scene, two):
    """+-3 % of measurement noise (a trial keeps the faster of two frames; a challenger must win by 3 %)."""
    r = sim(scene, 2, 8000, noise=0.06)
    assert r["final_two"] == two and abs(r["tail_two_share"] - two) <= 0.03, r
