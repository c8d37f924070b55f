#!/usr/bin/env python3
"""Generates the committed golden vectors (tests/golden/*.npz) from the CPU oracle.

The reference ships no golden data and cannot be executed in this image, so these vectors
pin the ORACLE'S behaviour at the time they were made (regression anchors for both the oracle and the HIP path),
not the reference's.  Each file holds inputs (60-float Splat records, the 32+3 camera floats, frame parameters)
and every stage's expected output.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from conftest import make_case, oracle_frame  # noqa: E402
from godotgaussiansplatting_amd import scenes  # noqa: E402

CASES = {
    # name: kwargs for conftest.make_case
    "tiny_deg3": dict(n=500, width=96, height=64, seed=101, sh_degree=3, scale_n=4000),
    # > 256 splats per tile and the block early-exit rule fires (composited < sorted)
    "tiny_dense_deg0": dict(n=1400, width=40, height=32, seed=102, sh_degree=0, scale_n=50,
                            camera=scenes.default_camera(3.0)),
    "tiny_anim_offaxis": dict(n=500, width=80, height=56, seed=103, sh_degree=1, scale_n=4000, model_scale=1.4,
                              time=0.8, load_time=0.0, heatmap=1.0,
                              camera=scenes.look_at_camera((2.5, 1.0, 4.0), target=(0.2, -0.1, 0.0))),
}


def main():
    for name, kw in CASES.items():
        case = make_case(**kw)
        gx = (case["width"] + 15) // 16
        target = 1 * gx + 2
        case["target_tile"] = target
        out = oracle.render_frame(case["records"], oracle_frame(case))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(
            path, records=case["records"], vp=case["vp"], cam_pos=case["cam_pos"],
            params=np.array([case["width"], case["height"], case["model_scale"], case["time"], case["heatmap"], target],
                            np.float64),
            culled=out["culled"], counts=out["counts"], keys_unsorted=out["keys_unsorted"],
            values_unsorted=out["values_unsorted"], keys=out["keys"], values=out["values"], bounds=out["bounds"],
            image=out["image"], pick=out["pick"],
            stats=np.array([out["stats"][k] for k in ("visible", "emitted", "sorted", "composited", "evals")], np.int64))
        print(name, os.path.getsize(path) // 1024, "KiB", out["stats"])


if __name__ == "__main__":
    main()
