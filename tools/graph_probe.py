"""Does a hipGraph of the frame's launches run faster than the launches themselves?  (static camera, one context)"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes, _lib
lib = _lib.load()
hip = C.CDLL(None)  # the HIP runtime already loaded by the library
for cfg, stripe in (("c1", None), ("c3", None), ("c3", (56, 64))):
    n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
    rows = scenes.synthetic_rows(n, seed, deg)
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    ctx = capi.Context(n, w, h, stream=s.value, stripe=(capi.STRIPE_COLUMNS,) + stripe if stripe else (0, 0, 0))
    for first in range(0, n, 1 << 20):
        ctx.upload_ply_rows(rows[first:first + (1 << 20)], first=first)
    fr = capi.make_frame(vp, cam)
    for _ in range(5):
        ctx.render(fr)
    ctx.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.render(fr)
    ctx.synchronize()
    plain = (time.perf_counter() - t0) / reps * 1e3
    graph, gexec = C.c_void_p(), C.c_void_p()
    rc = hip.hipStreamBeginCapture(s, 0)
    ctx.render(fr)
    rc2 = hip.hipStreamEndCapture(s, C.byref(graph))
    rc3 = hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0)
    if rc or rc2 or rc3:
        print(cfg, stripe, "capture failed", rc, rc2, rc3); continue
    for _ in range(5):
        hip.hipGraphLaunch(gexec, s)
    hip.hipStreamSynchronize(s)
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.hipGraphLaunch(gexec, s)
    hip.hipStreamSynchronize(s)
    g = (time.perf_counter() - t0) / reps * 1e3
    print(f"{cfg} stripe={stripe}: plain launches {plain:.4f} ms/frame, hipGraph {g:.4f} ms/frame")
    ctx.close()
