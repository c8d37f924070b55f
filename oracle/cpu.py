"""ctypes binding of oracle/gsplat_oracle.c (the C restatement of the reference pipeline).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None


class Frame(C.Structure):
    """gso_frame: per-frame inputs (uniform block + push constants of the reference)."""
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("cam_pos", C.c_float * 3),
                ("model_scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32), ("time", C.c_float),
                ("heatmap_factor", C.c_float), ("target_tile", C.c_uint32),
                ("stripe_x0", C.c_uint32), ("stripe_x1", C.c_uint32), ("stripe_y0", C.c_uint32),
                ("stripe_y1", C.c_uint32)]

    @classmethod
    def make(cls, view_proj32, cam_pos, width, height, model_scale=1.0, time=0.0, heatmap_factor=0.0,
             target_tile=0xFFFFFFFF, stripe=None):
        f = cls()
        vp = np.asarray(view_proj32, dtype=np.float32).reshape(32)
        f.view[:] = vp[:16].tolist()
        f.proj[:] = vp[16:].tolist()
        f.cam_pos[:] = np.asarray(cam_pos, dtype=np.float32).tolist()
        f.model_scale, f.width, f.height, f.time = model_scale, int(width), int(height), time
        f.heatmap_factor, f.target_tile = heatmap_factor, int(target_tile) & 0xFFFFFFFF
        sx0, sx1, sy0, sy1 = stripe if stripe is not None else (0, 0xFFFFFFFF, 0, 0xFFFFFFFF)
        f.stripe_x0, f.stripe_x1, f.stripe_y0, f.stripe_y1 = sx0, sx1, sy0, sy1
        return f


class Stats(C.Structure):
    _fields_ = [("visible", C.c_uint64), ("emitted", C.c_uint64), ("sorted", C.c_uint64),
                ("composited", C.c_uint64), ("evals", C.c_uint64), ("wave_steps", C.c_uint64),
                ("wave_full", C.c_uint64), ("overflow", C.c_int32),
                ("sig_bits", C.c_int32)]


def lib_path() -> str:
    return _build.SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    so = _build.build()
    try:
        lib = C.CDLL(so)
    except OSError:
        so = _build.build(force=True)  # shipped .so does not load on this host: rebuild in place
        lib = C.CDLL(so)
    u32p, f32p, u64 = C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_uint64
    lib.gso_project.restype = u64
    lib.gso_project.argtypes = [f32p, C.c_uint32, C.POINTER(Frame), u64, f32p, u32p, u32p, u32p, C.POINTER(u64), u32p]
    lib.gso_sort_pairs.argtypes = [u32p, u32p, u64]
    lib.gso_boundaries.argtypes = [u32p, u64, C.c_uint32, u32p, C.c_int, C.c_uint32]
    lib.gso_render.argtypes = [f32p, u32p, u32p, C.POINTER(Frame), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_float, f32p, f32p, C.POINTER(Stats)]
    lib.gso_frame_render.restype = C.c_int
    lib.gso_frame_render.argtypes = [f32p, C.c_uint32, C.POINTER(Frame), u64, f32p, u32p, u32p, u32p, u32p, u32p,
                                     u32p, f32p, f32p, C.POINTER(Stats)]
    lib.gso_ply_rows_to_records.argtypes = [f32p, C.c_uint32, C.c_float, f32p]
    lib.gso_pack_camera.argtypes = [f32p, f32p, f32p]
    lib.gso_pow02_array.argtypes = [f32p, f32p, u64]
    lib.gso_pow02_bits.argtypes = [C.c_uint32, f32p, u64]
    lib.gso_exp2_array.argtypes = [f32p, f32p, u64]
    lib.gso_num_threads.restype = C.c_int
    lib.gso_set_num_threads.argtypes = [C.c_int]
    _lib = lib
    return lib


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def grid(width, height):
    """tile_dims of gaussian_splatting_rasterizer.gd:29."""
    return (int(width) + 15) // 16, (int(height) + 15) // 16


def render_frame(records, frame, capacity=None, want_image=True):
    """rasterize() of gaussian_splatting_rasterizer.gd:122-160 on the CPU.  records: (N,60) f32."""
    lib = _load()
    rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 60)
    n = rec.shape[0]
    cap = int(capacity if capacity is not None else 10 * n)
    gx, gy = grid(frame.width, frame.height)
    out = {
        "culled": np.zeros((n, 12), np.float32), "counts": np.zeros(n, np.uint32),
        "keys": np.zeros(max(cap, 1), np.uint32), "values": np.zeros(max(cap, 1), np.uint32),
        "keys_unsorted": np.zeros(max(cap, 1), np.uint32), "values_unsorted": np.zeros(max(cap, 1), np.uint32),
        "bounds": np.zeros((gx * gy, 2), np.uint32), "pick": np.zeros(4, np.float32),
    }
    img = np.zeros((frame.height, frame.width, 4), np.float32) if want_image else None
    st = Stats()
    lib.gso_frame_render(_f32(rec), n, C.byref(frame), cap, _f32(out["culled"]), _u32(out["counts"]),
                         _u32(out["keys"]), _u32(out["values"]), _u32(out["keys_unsorted"]),
                         _u32(out["values_unsorted"]), _u32(out["bounds"]), _f32(img) if want_image else None,
                         _f32(out["pick"]), C.byref(st))
    d = int(st.sorted)
    for k in ("keys", "values", "keys_unsorted", "values_unsorted"):
        out[k] = out[k][:d]
    out["image"] = img
    out["stats"] = {f: int(getattr(st, f)) for f, _ in Stats._fields_}
    out["D"] = d
    return out


def project(records, frame, capacity=None):
    """gsplat_projection.glsl only: RasterizeData, per-splat tile counts, emission-order pairs."""
    lib = _load()
    rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 60)
    n = rec.shape[0]
    cap = int(capacity if capacity is not None else 10 * n)
    culled = np.zeros((n, 12), np.float32)
    counts = np.zeros(n, np.uint32)
    keys = np.zeros(max(cap, 1), np.uint32)
    values = np.zeros(max(cap, 1), np.uint32)
    vis = C.c_uint64(0)
    last = np.zeros(1, np.uint32)
    d_all = lib.gso_project(_f32(rec), n, C.byref(frame), cap, _f32(culled), _u32(counts), _u32(keys), _u32(values),
                            C.byref(vis), _u32(last))
    d = min(int(d_all), cap)
    return {"culled": culled, "counts": counts, "keys": keys[:d], "values": values[:d], "D": d,
            "emitted": int(d_all), "visible": int(vis.value), "frame_last_tile_plus1": int(last[0])}


def sort_pairs(keys, values):
    lib = _load()
    k = np.array(keys, dtype=np.uint32, copy=True)
    v = np.array(values, dtype=np.uint32, copy=True)
    lib.gso_sort_pairs(_u32(k), _u32(v), k.size)
    return k, v


def boundaries(sorted_keys, num_tiles, sharded=False, frame_last_tile_plus1=0):
    lib = _load()
    k = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    if k.size == 0:
        k = np.zeros(1, np.uint32)
        d = 0
    else:
        d = k.size
    b = np.zeros((num_tiles, 2), np.uint32)
    lib.gso_boundaries(_u32(k), d, num_tiles, _u32(b), int(bool(sharded)), int(frame_last_tile_plus1))
    return b


def render_tiles(culled, values, bounds, frame, exp_scale=1.0, tiles=None):
    """Compositor only (gsplat_render.glsl).  exp_scale != 1 perturbs every exp() (knife-edge detection)."""
    lib = _load()
    gx, gy = grid(frame.width, frame.height)
    x0, x1, y0, y1 = tiles if tiles is not None else (0, gx, 0, gy)
    img = np.zeros((frame.height, frame.width, 4), np.float32)
    pick = np.zeros(4, np.float32)
    st = Stats()
    c = np.ascontiguousarray(culled, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.uint32)
    if v.size == 0:
        v = np.zeros(1, np.uint32)
    b = np.ascontiguousarray(bounds, dtype=np.uint32)
    lib.gso_render(_f32(c), _u32(v), _u32(b), C.byref(frame), x0, x1, y0, y1, exp_scale, _f32(img), _f32(pick),
                   C.byref(st))
    return img, pick, {"composited": int(st.composited), "evals": int(st.evals), "wave_steps": int(st.wave_steps),
                       "wave_full": int(st.wave_full)}


def records_from_ply_rows(rows, load_time=-10.0):
    """ply_file.gd:41-69 — (N,62) INRIA rows -> (N,60) Splat records."""
    lib = _load()
    r = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, 62)
    out = np.zeros((r.shape[0], 60), np.float32)
    lib.gso_ply_rows_to_records(_f32(r), r.shape[0], load_time, _f32(out))
    return out


def pack_camera(cam12, proj16):
    """update_camera_matrices() packing, gaussian_splatting_rasterizer.gd:181-193."""
    lib = _load()
    cam = np.ascontiguousarray(cam12, dtype=np.float32).reshape(12)
    pr = np.ascontiguousarray(proj16, dtype=np.float32).reshape(16)
    out = np.zeros(32, np.float32)
    lib.gso_pack_camera(_f32(cam), _f32(pr), _f32(out))
    return out


def pow02(x):
    lib = _load()
    a = np.ascontiguousarray(x, dtype=np.float32).ravel()
    o = np.empty_like(a)
    lib.gso_pow02_array(_f32(a), _f32(o), a.size)
    return o


def pow02_bits(first_bits, count):
    """gso_pow02 of the `count` floats whose bit patterns start at first_bits."""
    lib = _load()
    o = np.empty(int(count), np.float32)
    lib.gso_pow02_bits(int(first_bits), _f32(o), int(count))
    return o


def exp2(x):
    lib = _load()
    a = np.ascontiguousarray(x, dtype=np.float32).ravel()
    o = np.empty_like(a)
    lib.gso_exp2_array(_f32(a), _f32(o), a.size)
    return o


def num_threads():
    return int(_load().gso_num_threads())


def set_num_threads(n):
    _load().gso_set_num_threads(int(n))
