#!/usr/bin/env python3
"""Two frames in flight on DISJOINT sets of compute units: each context on a stream made with hipExtStreamCreateWithCUMask.
Does giving each frame its own part of the chip beat letting the two frames' kernels share every CU (the default `value`)?
usage: python tools/cu_mask_probe.py <config>"""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from godotgaussiansplatting_amd import capi, scenes

# dlsym through libgsplat_hip.so's own handle: the symbol of the runtime copy THAT library is linked against (a process may
# hold two copies of libamdhip64 — torch's and /opt/rocm's — and a stream of one is nothing to the other)
hip = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "godotgaussiansplatting_amd", "libgsplat_hip.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(bits):   # bits: list of 256 0/1
    words = (ctypes.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    hip.hipSetDevice(0)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s.value


cfg = sys.argv[1]
n, deg, w, h, seed, vp, cam = bench.build_scene_inputs(cfg)
rows = scenes.config_rows(cfg)
fr = capi.make_frame(vp, cam)
base = capi.Context(n, w, h)
for first in range(0, n, 1 << 20):
    base.upload_ply_rows(rows[first:first + (1 << 20)], first=first, load_time=-10.0)
base.render(fr); base.synchronize()


def rate(ring, reps=300):
    for c in ring:
        for _ in range(64):
            c.render(fr); c.synchronize()
    for k in range(20):
        ring[k % len(ring)].render(fr)
    for c in ring:
        c.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        ring[k % len(ring)].render(fr)
    for c in ring:
        c.synchronize()
    return reps / (time.perf_counter() - t0)


print(cfg, "default streams: 1 in flight %.0f, 2 in flight %.0f" % (rate([base]), rate([base, base.view()])))
LAYOUTS = {
    "halves (bits 0-127 | 128-255)": ([1 if i < 128 else 0 for i in range(256)], [0 if i < 128 else 1 for i in range(256)]),
    "alternating bits (even | odd)": ([1 - (i & 1) for i in range(256)], [i & 1 for i in range(256)]),
    "alternating groups of 8 bits": ([1 - ((i >> 3) & 1) for i in range(256)], [(i >> 3) & 1 for i in range(256)]),
    "alternating groups of 32 bits": ([1 - ((i >> 5) & 1) for i in range(256)], [(i >> 5) & 1 for i in range(256)]),
}
for name, (ma, mb) in LAYOUTS.items():
    sa, sb = masked_stream(ma), masked_stream(mb)
    a = base.view(stream=sa)
    b = base.view(stream=sb)
    one = rate([a])
    two = rate([a, b])
    print(f"{cfg} {name}: one context on its half {one:.0f} frames/s; two contexts, one per half {two:.0f}")
    a.close(); b.close()
full = masked_stream([1] * 256)
a = base.view(stream=full)
print(cfg, "all 256 bits set: %.0f" % rate([a]))
