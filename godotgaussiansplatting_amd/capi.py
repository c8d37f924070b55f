"""Thin object wrapper over the C ABI (include/gsplat.h) — one method per entry point, NumPy in/out.
Used by the parity tests and bench.py; GaussianSplattingRasterizer is the reference-shaped interface."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FLAG_BLOCK_CULL, FLAG_TIES_STORAGE_ORDER, FLAG_READBACK_RGB, FLAG_FAST_EXP, FLAG_FIX_LAST_TILE, FLAG_KEEP_EMITTED, FLAG_KERNEL_TIMING, FLAG_TIMING, NO_TARGET_TILE,  # noqa: F401
                   STRIPE_COLUMNS, STRIPE_NONE, STRIPE_ROWS)


def make_view_proj(camera_xform12, fov, aspect, near, far, basis_override9=None):
    """gsplat_make_view_proj -> (view|proj 32 floats, uniform cam_pos 3 floats)."""
    lib = _lib.load()
    f32p = C.POINTER(C.c_float)
    cam = np.ascontiguousarray(camera_xform12, np.float32).reshape(12)
    bo = None if basis_override9 is None else np.ascontiguousarray(basis_override9, np.float32).reshape(9)
    out32, pos = np.zeros(32, np.float32), np.zeros(3, np.float32)
    _lib.check(lib.gsplat_make_view_proj(cam.ctypes.data_as(f32p), bo.ctypes.data_as(f32p) if bo is not None else None,
                                         fov, aspect, near, far, out32.ctypes.data_as(f32p), pos.ctypes.data_as(f32p)),
               "gsplat_make_view_proj")
    return out32, pos


def device_count():
    """Number of HIP devices this process sees, asked of the HIP runtime the library itself runs on (no torch)."""
    _lib.load()
    n = C.c_int(0)
    for handle in (None, "libamdhip64.so", "libamdhip64.so.7"):
        try:
            fn = C.CDLL(handle).hipGetDeviceCount
        except (OSError, AttributeError):
            continue
        fn.argtypes = [C.POINTER(C.c_int)]
        return int(n.value) if fn(C.byref(n)) == 0 else 0
    return 0


def make_frame(view_proj32, cam_pos, model_scale=1.0, time=0.0, heatmap_factor=0.0, target_tile=NO_TARGET_TILE):
    f = _lib.Frame()
    vp = np.asarray(view_proj32, np.float32).reshape(32)
    f.view[:] = vp[:16].tolist()
    f.proj[:] = vp[16:].tolist()
    f.cam_pos[:] = np.asarray(cam_pos, np.float32).tolist()
    f.model_scale, f.time, f.heatmap_factor = model_scale, time, heatmap_factor
    f.target_tile = int(target_tile) & 0xFFFFFFFF
    return f


class Context:
    def __init__(self, max_splats, width, height, *, key_budget_factor=10, device_id=-1, flags=0,
                 stripe=(STRIPE_NONE, 0, 0), sh_degree=-1, stream=None, scene_of=None, batch=1):
        """scene_of: another Context — the new one renders that context's scene (gsplat_create_view: own size, stripe,
        stream and intermediate buffers, one shared splat buffer).  batch > 1 (with scene_of): gsplat_create_batch_view —
        the intermediate buffers hold `batch` frames, rendered through one launch sequence by render_batch."""
        self.lib = _lib.load()
        cfg = _lib.Config()
        cfg.struct_size = C.sizeof(_lib.Config)
        cfg.max_splats, cfg.width, cfg.height = int(max_splats), int(width), int(height)
        cfg.key_budget_factor, cfg.device_id, cfg.flags = int(key_budget_factor), int(device_id), int(flags)
        cfg.stripe_axis, cfg.stripe_begin, cfg.stripe_end = stripe
        cfg.sh_degree = sh_degree
        cfg.stream = stream
        self.ctx = C.c_void_p()
        self.batch = int(batch)
        if scene_of is None:
            assert self.batch == 1, "a batch context is a view on a scene owner (Context.view(batch=B))"
            _lib.check(self.lib.gsplat_create(C.byref(cfg), C.byref(self.ctx)), "gsplat_create")
        elif self.batch > 1:
            _lib.check(self.lib.gsplat_create_batch_view(scene_of.ctx, C.byref(cfg), self.batch, C.byref(self.ctx)),
                       "gsplat_create_batch_view")
            max_splats = scene_of.n
        else:
            _lib.check(self.lib.gsplat_create_view(scene_of.ctx, C.byref(cfg), C.byref(self.ctx)), "gsplat_create_view")
            max_splats = scene_of.n
        self.n, self.width, self.height = int(max_splats), int(width), int(height)
        self.readback_channels = 3 if int(flags) & FLAG_READBACK_RGB else 4

    def view(self, width=None, height=None, **kwargs):
        """A second context on this context's scene."""
        return Context(self.n, width or self.width, height or self.height, scene_of=self, **kwargs)

    @property
    def tiles(self):
        return ((self.width + 15) // 16) * ((self.height + 15) // 16)

    def close(self):
        if self.ctx:
            # (refused while the context is a member of a gsplat_group: the handle stays valid, close the group first)
            _lib.check(self.lib.gsplat_destroy(self.ctx), "gsplat_destroy")
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_splats(self, records, first=0):
        r = np.ascontiguousarray(records, np.float32).reshape(-1, 60)
        _lib.check(self.lib.gsplat_upload_splats(self.ctx, first, r.shape[0], r.ctypes.data_as(C.c_void_p)),
                   "gsplat_upload_splats")

    def upload_ply_rows(self, rows, first=0, load_time=-10.0):
        r = np.ascontiguousarray(rows, np.float32).reshape(-1, 62)
        _lib.check(self.lib.gsplat_upload_ply_rows(self.ctx, first, r.shape[0], r.ctypes.data_as(C.c_void_p),
                                                   C.c_float(load_time)), "gsplat_upload_ply_rows")

    def finalize_scene(self):
        _lib.check(self.lib.gsplat_finalize_scene(self.ctx), "gsplat_finalize_scene")

    def resize(self, width, height):
        _lib.check(self.lib.gsplat_resize(self.ctx, int(width), int(height)), "gsplat_resize")
        self.width, self.height = int(width), int(height)

    def set_stripe(self, axis, begin, end):
        _lib.check(self.lib.gsplat_set_stripe(self.ctx, axis, begin, end), "gsplat_set_stripe")

    def render(self, frame, out=None):
        """out: None (stay on device), a host ndarray (H,W,4) f32, or an int device pointer."""
        ptr = None
        if out is not None:
            ptr = C.c_void_p(out) if isinstance(out, int) else out.ctypes.data_as(C.c_void_p)
        _lib.check(self.lib.gsplat_render(self.ctx, C.byref(frame), ptr), "gsplat_render")

    @staticmethod
    def _frame_array(frames):
        arr = (_lib.Frame * len(frames))()
        for k, f in enumerate(frames):
            C.memmove(C.byref(arr, k * C.sizeof(_lib.Frame)), C.byref(f), C.sizeof(_lib.Frame))
        return arr

    def render_batch(self, frames):
        """gsplat_render_batch: len(frames) <= batch frames through one launch sequence; frame k -> batch image k."""
        _lib.check(self.lib.gsplat_render_batch(self.ctx, self._frame_array(frames), len(frames)), "gsplat_render_batch")

    def render_batch_begin(self, frames, last_tiles_out_ptr=None):
        _lib.check(self.lib.gsplat_render_batch_begin(self.ctx, self._frame_array(frames), len(frames),
                                                      C.c_void_p(int(last_tiles_out_ptr) if last_tiles_out_ptr else None)),
                   "gsplat_render_batch_begin")

    def render_batch_end(self, frame_last_tiles_ptr=None):
        _lib.check(self.lib.gsplat_render_batch_end(self.ctx, C.c_void_p(int(frame_last_tiles_ptr) if frame_last_tiles_ptr else None)),
                   "gsplat_render_batch_end")

    def read_batch_images(self, count):
        """The images of the last batch: (count, H, W, 4) float32 (GSPLAT_DEBUG_IMAGE after gsplat_render_batch)."""
        return self.debug_read(_lib.DEBUG_IMAGE, np.float32, count * self.width * self.height * 4).reshape(
            count, self.height, self.width, 4)

    def render_to_host(self, frame):
        img = np.empty((self.height, self.width, 4), np.float32)
        self.render(frame, img)
        return img

    def render_async(self, frame):
        """gsplat_render_async: the frame goes to a pinned host image behind the scenes; returns its ticket."""
        t = C.c_uint64(0)
        _lib.check(self.lib.gsplat_render_async(self.ctx, C.byref(frame), C.byref(t)), "gsplat_render_async")
        return t.value

    def readback_wait(self, ticket):
        """gsplat_readback_wait -> (H, W, 4) float32 VIEW of the library's pinned image (valid for two more frames);
        (H, W, 3) for a context created with FLAG_READBACK_RGB."""
        p = C.POINTER(C.c_float)()
        _lib.check(self.lib.gsplat_readback_wait(self.ctx, C.c_uint64(ticket), C.byref(p)), "gsplat_readback_wait")
        return np.ctypeslib.as_array(p, shape=(self.height, self.width, self.readback_channels))

    def export_image_fd(self):
        fd, size = C.c_int(-1), C.c_uint64(0)
        _lib.check(self.lib.gsplat_export_image_fd(self.ctx, C.byref(fd), C.byref(size)), "gsplat_export_image_fd")
        return fd.value, size.value

    def bind_external_image(self, fd, size_bytes, offset_bytes=0):
        _lib.check(self.lib.gsplat_bind_external_image(self.ctx, int(fd), int(size_bytes), int(offset_bytes)),
                   "gsplat_bind_external_image")

    def pick(self, frame, tile_id):
        out = (C.c_float * 4)()
        _lib.check(self.lib.gsplat_pick(self.ctx, C.byref(frame), int(tile_id), out), "gsplat_pick")
        return np.array(list(out), np.float32)

    def synchronize(self):
        _lib.check(self.lib.gsplat_synchronize(self.ctx), "gsplat_synchronize")

    def stats(self):
        st = _lib.Stats()
        st.struct_size = C.sizeof(_lib.Stats)
        _lib.check(self.lib.gsplat_get_stats(self.ctx, C.byref(st)), "gsplat_get_stats")
        skip = ("algorithmic_bytes", "ms_kernel", "launches_kernel", "pairs_round", "struct_size", "reserved0")
        d = {name: getattr(st, name) for name, _ in _lib.Stats._fields_ if name not in skip}
        d["algorithmic_bytes"] = [int(x) for x in st.algorithmic_bytes]
        d["pairs_round"] = [int(x) for x in st.pairs_round]
        d["ms_kernel"] = {k: float(st.ms_kernel[i]) for i, k in enumerate(_lib.KERNEL_CLASSES)}
        d["launches_kernel"] = {k: int(st.launches_kernel[i]) for i, k in enumerate(_lib.KERNEL_CLASSES)}
        return d

    def set_timing(self, flags):
        _lib.check(self.lib.gsplat_set_timing(self.ctx, int(flags)), "gsplat_set_timing")

    def render_to(self, frame, device_ptr, pitch_px, origin_x=0, origin_y=0):
        _lib.check(self.lib.gsplat_render_to(self.ctx, C.byref(frame), C.c_void_p(int(device_ptr)), int(pitch_px),
                                             int(origin_x), int(origin_y)), "gsplat_render_to")

    def render_begin(self, frame, last_tile_out_ptr=None):
        """gsplat_render_begin: projection, emission, sort; the context's own 'highest populated tile + 1' goes to the
        4 device bytes at last_tile_out_ptr (None to skip)."""
        _lib.check(self.lib.gsplat_render_begin(self.ctx, C.byref(frame),
                                                C.c_void_p(int(last_tile_out_ptr) if last_tile_out_ptr else None)),
                   "gsplat_render_begin")

    def render_end(self, device_ptr=None, pitch_px=0, origin_x=0, origin_y=0, frame_last_tile_ptr=None):
        """gsplat_render_end: tile ranges + compositor; frame_last_tile_ptr = device word with the MAX over all ranks
        of what render_begin reported (None: this context's own value)."""
        _lib.check(self.lib.gsplat_render_end(self.ctx, C.c_void_p(int(device_ptr) if device_ptr else None),
                                              int(pitch_px), int(origin_x), int(origin_y),
                                              C.c_void_p(int(frame_last_tile_ptr) if frame_last_tile_ptr else None)),
                   "gsplat_render_end")

    def image_device_ptr(self):
        p = C.c_void_p()
        _lib.check(self.lib.gsplat_image_device_ptr(self.ctx, C.byref(p)), "gsplat_image_device_ptr")
        return p.value

    def debug_read(self, which, dtype, count):
        buf = np.zeros(int(count), dtype)
        n = C.c_size_t(0)
        _lib.check(self.lib.gsplat_debug_read(self.ctx, which, buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(n)),
                   "gsplat_debug_read")
        return buf[: n.value // buf.itemsize]

    def debug_pow02(self, first_bits, count):
        """The kernels' pow(x, 0.2) for the floats with bit patterns first_bits .. first_bits + count - 1."""
        out = np.empty(int(count), np.float32)
        _lib.check(self.lib.gsplat_debug_pow02(self.ctx, int(first_bits), int(count), out.ctypes.data_as(C.POINTER(C.c_float))),
                   "gsplat_debug_pow02")
        return out

    def sort_rank_mode(self):
        """'atomic' or 'ballot': how this context's sort downsweeps rank (GSPLAT_DEBUG_SORT_RANK)."""
        return "atomic" if int(self.debug_read(_lib.DEBUG_SORT_RANK, np.uint32, 1)[0]) else "ballot"

    # convenience taps
    def emit_lists_big_rectangles(self):
        """True: the last frame listed its rectangles of more than 512 tiles for emit_big_kernel (DEBUG_EMIT_MODE)."""
        return bool(self.debug_read(_lib.DEBUG_EMIT_MODE, np.uint32, 1)[0])

    def read_culled(self):
        return self.debug_read(_lib.DEBUG_CULLED, np.float32, self.n * 12).reshape(-1, 12)

    def read_counts(self):
        return self.debug_read(_lib.DEBUG_TILE_COUNTS, np.uint32, self.n)

    def read_sorted(self):
        d = self.stats()["num_sorted"]
        return (self.debug_read(_lib.DEBUG_KEYS_SORTED, np.uint32, d), self.debug_read(_lib.DEBUG_VALUES_SORTED, np.uint32, d))

    def read_emitted(self):
        d = self.stats()["num_sorted"]
        return (self.debug_read(_lib.DEBUG_KEYS_EMITTED, np.uint32, d), self.debug_read(_lib.DEBUG_VALUES_EMITTED, np.uint32, d))

    def read_bounds(self):
        return self.debug_read(_lib.DEBUG_TILE_BOUNDS, np.uint32, self.tiles * 2).reshape(-1, 2)

    def read_records(self):
        return self.debug_read(_lib.DEBUG_RECORDS, np.float32, self.n * 60).reshape(-1, 60)

    def read_tile_staged(self):
        return self.debug_read(_lib.DEBUG_TILE_STAGED, np.uint32, self.tiles)

    def read_tile_order(self, stripe_tiles=None):
        """The compositor's schedule of the last frame: tile ids of the stripe, most expensive first."""
        return self.debug_read(_lib.DEBUG_TILE_ORDER, np.uint32, self.tiles if stripe_tiles is None else stripe_tiles)

    def read_block_sums(self):
        """(ceil(N/512), 4) uint32 per projection workgroup: pairs, visible, last tile + 1, skipped-by-block-cull."""
        nb = (self.n + 511) // 512
        return self.debug_read(_lib.DEBUG_BLOCK_SUMS, np.uint32, nb * 4).reshape(nb, 4)

    def read_slot_ids(self):
        """id_of_slot: the splat id stored in each slot (identity until finalize_scene)."""
        return self.debug_read(_lib.DEBUG_SLOT_IDS, np.uint32, self.n)

    def read_image(self):
        return self.debug_read(_lib.DEBUG_IMAGE, np.float32, self.width * self.height * 4).reshape(self.height, self.width, 4)


GROUP_ID_BYTES = 128


def group_unique_id():
    """gsplat_group_unique_id: rank 0 makes it, the other ranks receive the 128 bytes over any channel."""
    _lib.share_rccl_with_torch()
    lib = _lib.load()
    buf = (C.c_uint8 * GROUP_ID_BYTES)()
    _lib.check(lib.gsplat_group_unique_id(buf), "gsplat_group_unique_id")
    return bytes(buf)


class Group:
    """gsplat_group_*: one frame sharded by tile stripes over the GPUs of a node, the exchange inside the library (RCCL
    on the members' streams).  Group(ctx, unique_id, rank, world) in the one-process-per-GPU form;
    Group.local([ctx0, ctx1, ...]) in the one-process form (one context per device)."""

    def __init__(self, ctx, unique_id, rank, world, axis=STRIPE_ROWS):
        _lib.share_rccl_with_torch()
        self.lib = _lib.load()
        self.members = [ctx]
        self.world = int(world)
        self.group = C.c_void_p()
        idb = (C.c_uint8 * GROUP_ID_BYTES).from_buffer_copy(unique_id)
        _lib.check(self.lib.gsplat_group_create(ctx.ctx, idb, int(rank), int(world), int(axis), C.byref(self.group)),
                   "gsplat_group_create")

    @classmethod
    def local(cls, ctxs, axis=STRIPE_ROWS):
        _lib.share_rccl_with_torch()
        self = cls.__new__(cls)
        self.lib = _lib.load()
        self.members = list(ctxs)
        self.world = len(self.members)
        self.group = C.c_void_p()
        arr = (C.c_void_p * self.world)(*[c.ctx for c in self.members])
        _lib.check(self.lib.gsplat_group_create_local(arr, self.world, int(axis), C.byref(self.group)),
                   "gsplat_group_create_local")
        return self

    def set_cuts(self, cuts):
        arr = (C.c_uint32 * (self.world + 1))(*[int(c) for c in cuts])
        _lib.check(self.lib.gsplat_group_set_cuts(self.group, arr), "gsplat_group_set_cuts")

    def render_batch(self, frames):
        """gsplat_group_render_batch: len(frames) consecutive frames of every local member (batch contexts) through one
        launch sequence, ONE all-reduce of that many words, ONE all-gather-v of that many stripes."""
        _lib.check(self.lib.gsplat_group_render_batch(self.group, Context._frame_array(frames), len(frames)),
                   "gsplat_group_render_batch")

    def render(self, frame, outs=None):
        arr = None
        if outs is not None:
            arr = (C.c_void_p * len(self.members))(*[int(o) if o else None for o in outs])
        _lib.check(self.lib.gsplat_group_render(self.group, C.byref(frame), arr), "gsplat_group_render")

    def exchanges_last_tile(self):
        """True: every frame of this group carries the 4-byte all-reduce (agreed by all ranks at creation)."""
        return bool(_lib.check_value(self.lib.gsplat_group_exchanges_last_tile(self.group), "gsplat_group_exchanges_last_tile"))

    def close(self):
        if self.group:
            self.lib.gsplat_group_destroy(self.group)
            self.group = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
