import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that does not come back (a kernel that never ends blocks hipStreamSynchronize inside a C call, where no
    signal handler runs) must end the run, not hold the box until somebody else's limit: the slowest GPU test takes
    ~25 s, so after 10 minutes pytest-timeout's watchdog thread dumps the stacks and exits the process."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


def godot_perspective(fov_deg, aspect, near, far):
    """Godot 4.3 Projection::set_perspective in float32 (columns, 16 floats) — test-side restatement."""
    f = np.float32
    radians = f(f(fov_deg) / f(2.0)) * f(0.017453292519943295)
    sine = f(np.sin(radians, dtype=np.float32))
    cot = f(np.cos(radians, dtype=np.float32)) / sine
    dz = f(far) - f(near)
    p = np.zeros(16, np.float32)
    p[0] = cot / f(aspect)
    p[5] = cot
    p[10] = -(f(far) + f(near)) / dz
    p[11] = -1.0
    p[14] = f(-2.0) * f(near) * f(far) / dz
    return p


def make_case(n, width, height, seed=1, sh_degree=0, camera=None, model_scale=1.0, time=0.0, load_time=-10.0,
              heatmap=0.0, target_tile=0xFFFFFFFF, scale_n=None):
    """Scene + frame inputs shared by the oracle and the HIP path (same 60-float records, same 32+3 floats)."""
    import oracle
    from godotgaussiansplatting_amd import scenes
    rows = scenes.synthetic_rows(n, seed, sh_degree, scale_n=scale_n)
    records = oracle.records_from_ply_rows(rows, load_time)
    cam = camera or scenes.default_camera()
    proj = godot_perspective(cam.fov, width / height, cam.near, cam.far)
    vp = oracle.pack_camera(cam.xform12(), proj)
    cam_pos = np.array([-cam.origin[0], -cam.origin[1], cam.origin[2]], np.float32)
    return {"rows": rows, "records": records, "vp": vp, "cam_pos": cam_pos, "width": width, "height": height,
            "model_scale": model_scale, "time": time, "heatmap": heatmap, "target_tile": target_tile,
            "load_time": load_time, "camera": cam}


def oracle_frame(case, stripe=None):
    import oracle
    return oracle.Frame.make(case["vp"], case["cam_pos"], case["width"], case["height"], case["model_scale"],
                             case["time"], case["heatmap"], case["target_tile"], stripe)


def hip_frame(case):
    from godotgaussiansplatting_amd import capi
    return capi.make_frame(case["vp"], case["cam_pos"], case["model_scale"], case["time"], case["heatmap"],
                           case["target_tile"])


@pytest.fixture(scope="session")
def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
