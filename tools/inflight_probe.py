#!/usr/bin/env python3
"""Why did tools/stripe_batch.py see no gain from a second frame in flight where tools/stripe_model.py sees one?
Variants of the same ring loop on the c3 middle stripe (columns 55:60), printed side by side."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from godotgaussiansplatting_amd import capi, scenes
VARIANT = sys.argv[1]
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs("c3")
ROWS = scenes.config_rows("c3")
FLAGS = capi.FLAG_BLOCK_CULL | capi.FLAG_TIES_STORAGE_ORDER
if "torch_first" in VARIANT:
    import torch
    _t = torch.zeros(1, device="cuda")
ctx = capi.Context(n, w, h, flags=capi.FLAG_TIMING | FLAGS)
for first in range(0, n, 1 << 20):
    ctx.upload_ply_rows(ROWS[first:first + (1 << 20)], first=first, load_time=-10.0)
ctx.finalize_scene()
fr = capi.make_frame(vp, cam)
ctx.render(fr); ctx.synchronize()
if "notorch" in VARIANT:
    from godotgaussiansplatting_amd import _lib as _gl
    _gl.load()
    hip = ctypes.CDLL(None)
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(16)) == 0
    host = (ctypes.c_uint32 * 4)(8160, 8160, 8160, 8160)
    TOPP = p.value
    if "nonull" in VARIANT:   # never touch the null stream: the context's own frame writes the word
        ctx.render_begin(fr, TOPP); ctx.render_end(); ctx.synchronize()
    else:
        assert hip.hipMemcpy(p, host, ctypes.c_size_t(16), 1) == 0
    DUMMY = []
    for _k in range(int(os.environ.get("PROBE_DUMMY_STREAMS", "0"))):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        DUMMY.append(st)
        if "dummyuse" in VARIANT:
            assert hip.hipMemcpyAsync(p, host, ctypes.c_size_t(16), 1, st) == 0
            assert hip.hipStreamSynchronize(st) == 0
else:
    import torch
    TOP = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.render_begin(fr, TOP.data_ptr()); ctx.render_end(); ctx.synchronize()
    TOPP = TOP.data_ptr()


def render(c):
    c.render_begin(fr)
    c.render_end(frame_last_tile_ptr=TOPP)


for R in (1, 2, 3, 4):
    ring = [ctx.view(stripe=(capi.STRIPE_COLUMNS, 55, 60), flags=FLAGS) for _ in range(R)]
    for k in range(3 * R):
        render(ring[k % R])
    for c in ring:
        c.synchronize()
    reps = 60
    t0 = time.perf_counter()
    for k in range(reps):
        render(ring[k % R])
    for c in ring:
        c.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"{VARIANT}: {R} in flight {dt:.3f} ms/frame", flush=True)
    for c in ring:
        c.close()
ctx.close()
