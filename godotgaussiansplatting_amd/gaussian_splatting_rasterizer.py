"""GaussianSplattingRasterizer — host-side mirror of util/gaussian_splatting_rasterizer.gd.

Same names, argument meaning and behaviour as the reference's GDScript Resource class; the body of every
method calls libgsplat_hip.so through the C ABI of include/gsplat.h instead of recording Vulkan compute
lists.  There is no CPU fallback: construction works without a GPU (like the reference before its first
rasterize()), init_gpu() raises if the HIP library or a device is missing.
"""
import ctypes as C
import threading
import time

import numpy as np

from . import _lib
from .godot_types import Basis, Camera3D, Texture2DRD
from .ply_file import PlyFile


class Signal:
    """Godot signal stand-in (`signal loaded`, gaussian_splatting_rasterizer.gd:10)."""

    def __init__(self):
        self._slots = []

    def connect(self, fn):
        self._slots.append(fn)

    def emit(self, *a):
        for fn in list(self._slots):
            fn(*a)


class _UploadDevice:
    """What PlyFile.load_gaussian_splats sees as `device` (RenderingDevice.buffer_update, ply_file.gd:71)."""

    def __init__(self, lib, ctx):
        self.lib, self.ctx = lib, ctx

    def buffer_update_ply_rows(self, buffer, first, rows, creation_time):
        r = np.ascontiguousarray(rows, dtype=np.float32)
        _lib.check(self.lib.gsplat_upload_ply_rows(self.ctx, first, r.shape[0], r.ctypes.data_as(C.c_void_p),
                                                   C.c_float(creation_time)), "gsplat_upload_ply_rows")


class GaussianSplattingRasterizer:
    TILE_SIZE = 16          # gaussian_splatting_rasterizer.gd:4
    WORKGROUP_SIZE = 512    # :5 (reference sort constants, kept for interface parity)
    RADIX = 256             # :6
    PARTITION_DIVISION = 8  # :7
    PARTITION_SIZE = PARTITION_DIVISION * WORKGROUP_SIZE

    def __init__(self, point_cloud: PlyFile, output_texture_size, render_texture: Texture2DRD, camera: Camera3D, *,
                 device_id: int = -1, flags: int = 0, sh_degree: int = -1, stream=None, key_budget_factor: int = 10,
                 time_source=None, async_load: bool = True, finalize_when_loaded: bool = False):
        self.loaded = Signal()
        self.context = None                      # gsplat_ctx* (the reference's RenderingContext)
        self._lib = None
        self.load_thread = None
        self.is_loaded = False
        self.should_enable_heatmap = [False]
        self.render_scale = [1.0]
        self.model_scale = [1.0]
        self.should_terminate_thread = [False]
        self.num_splats_loaded = [0]
        self.basis_override = Basis()
        self.camera_push_constants = None        # 32 floats: view | proj
        self._camera_key = None
        self._uniform_cam_pos = np.zeros(3, np.float32)
        self.tile_dims = (0, 0)
        self._texture_size = (1, 1)
        self._opts = dict(device_id=device_id, flags=flags, sh_degree=sh_degree, stream=stream,
                          key_budget_factor=key_budget_factor)
        self._t0 = time.monotonic()
        self._time_source = time_source if time_source is not None else (lambda: time.monotonic() - self._t0)
        self._async_load = async_load
        self._last_frame = None
        # gsplat_finalize_scene once the loader has finished (the `loaded` signal): Morton re-layout of the stored
        # scene; pays for tile-stripe shards and cameras inside the scene, costs ~0.1 ms per frame on one GPU
        self._finalize_when_loaded = finalize_when_loaded
        self._finalized = False
        # gaussian_splatting_rasterizer.gd:59-63
        self.point_cloud = point_cloud
        self.texture_size = output_texture_size
        self.render_texture = render_texture
        self.camera = camera

    # -- texture_size property (gaussian_splatting_rasterizer.gd:26-48) ---------------------------------
    @property
    def texture_size(self):
        return self._texture_size

    @texture_size.setter
    def texture_size(self, value):
        w = max(1, int(value[0] * self.render_scale[0]))
        h = max(1, int(value[1] * self.render_scale[0]))
        self._texture_size = (w, h)
        self.tile_dims = ((w + self.TILE_SIZE - 1) // self.TILE_SIZE, (h + self.TILE_SIZE - 1) // self.TILE_SIZE)
        if not self.context:
            return
        # rebuild tile_bounds and the render texture (they depend on the texture size)
        _lib.check(self._lib.gsplat_resize(self.context, w, h), "gsplat_resize")
        self._bind_texture()

    def _bind_texture(self):
        ptr = C.c_void_p()
        _lib.check(self._lib.gsplat_image_device_ptr(self.context, C.byref(ptr)), "gsplat_image_device_ptr")
        self.render_texture.texture_rd_rid = ptr.value or 0
        self.render_texture.size = self._texture_size
        self.render_texture._reader = self._read_image

    def _read_image(self):
        w, h = self._texture_size
        img = np.empty((h, w, 4), np.float32)
        if self._last_frame is None:
            raise RuntimeError("nothing rendered yet")
        n = C.c_size_t(0)
        _lib.check(self._lib.gsplat_debug_read(self.context, _lib.DEBUG_IMAGE, img.ctypes.data_as(C.c_void_p),
                                               img.nbytes, C.byref(n)), "gsplat_debug_read")
        return img

    # -- init_gpu / cleanup_gpu (gaussian_splatting_rasterizer.gd:65-120) --------------------------------
    def init_gpu(self) -> None:
        assert self.render_texture is not None, "An output Texture2DRD must be set!"
        self._lib = _lib.load()
        cfg = _lib.Config()
        cfg.struct_size = C.sizeof(_lib.Config)
        cfg.max_splats = self.point_cloud.size
        cfg.width, cfg.height = self._texture_size
        cfg.key_budget_factor = self._opts["key_budget_factor"]   # point_cloud.size * 10, :79
        cfg.device_id = self._opts["device_id"]
        cfg.flags = self._opts["flags"]
        cfg.stripe_axis = _lib.STRIPE_NONE
        cfg.sh_degree = self._opts["sh_degree"]
        cfg.stream = self._opts["stream"]
        ctx = C.c_void_p()
        _lib.check(self._lib.gsplat_create(C.byref(cfg), C.byref(ctx)), "gsplat_create")
        self.context = ctx
        self._bind_texture()
        # begin loading splats asynchronously (:111-114)
        self.should_terminate_thread[0] = False
        self.num_splats_loaded[0] = 0
        args = (self.point_cloud, self.point_cloud.size // 1000, _UploadDevice(self._lib, self.context), None,
                self.should_terminate_thread, self.num_splats_loaded, self.loaded.emit, self._time_source)
        if self._async_load:
            self.load_thread = threading.Thread(target=PlyFile.load_gaussian_splats, args=args, daemon=True)
            self.load_thread.start()
        else:
            self.load_thread = None
            PlyFile.load_gaussian_splats(*args)

    def cleanup_gpu(self) -> None:
        self.should_terminate_thread[0] = True
        if self.load_thread is not None:
            self.load_thread.join()
            self.load_thread = None
        if self.context:
            self._lib.gsplat_destroy(self.context)
            self.context = None
        if self.render_texture is not None:
            self.render_texture.texture_rd_rid = 0
            self.render_texture._reader = None

    def __del__(self):
        try:
            self.cleanup_gpu()
        except Exception:
            pass

    # -- per-frame (gaussian_splatting_rasterizer.gd:122-160) ---------------------------------------------
    def _make_frame(self, target_tile=_lib.NO_TARGET_TILE):
        if self.camera_push_constants is None:
            self.update_camera_matrices()
        f = _lib.Frame()
        f.view[:] = self.camera_push_constants[:16].tolist()
        f.proj[:] = self.camera_push_constants[16:].tolist()
        # :125-126 camera_pos = basis_override * camera.global_position, uploaded as (-x, -y, z)
        f.cam_pos[:] = self._uniform_cam_pos.tolist()
        f.model_scale = float(self.model_scale[0])
        f.time = float(self._time_source())
        f.heatmap_factor = float(self.should_enable_heatmap[0])
        f.target_tile = int(target_tile) & 0xFFFFFFFF
        return f

    def rasterize(self, rgba_out=None) -> None:
        """Render one frame into render_texture.  rgba_out: optional device pointer (int) or host numpy
        array that receives the frame instead (C ABI rgba_out)."""
        if not self.context:
            self.init_gpu()
        self.update_camera_matrices()
        self.is_loaded = self.load_thread is None or not self.load_thread.is_alive()
        if self.is_loaded and self._finalize_when_loaded and not self._finalized:
            _lib.check(self._lib.gsplat_finalize_scene(self.context), "gsplat_finalize_scene")
            self._finalized = True
        frame = self._make_frame()
        out = None
        if rgba_out is not None:
            out = C.c_void_p(rgba_out) if isinstance(rgba_out, int) else rgba_out.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.gsplat_render(self.context, C.byref(frame), out), "gsplat_render")
        self._last_frame = frame

    def get_splat_position(self, screen_position):
        """gaussian_splatting_rasterizer.gd:162-171.  Returns a length-3 array, or all-inf if the tile is empty."""
        if self._last_frame is None:
            raise RuntimeError("get_splat_position needs a rendered frame")
        tx = int(screen_position[0] * self.render_scale[0] / self.TILE_SIZE)
        ty = int(screen_position[1] * self.render_scale[0] / self.TILE_SIZE)
        tile_id = ty * self.tile_dims[0] + tx
        out = (C.c_float * 4)()
        frame = self._last_frame
        _lib.check(self._lib.gsplat_pick(self.context, C.byref(frame), tile_id, out), "gsplat_pick")
        if out[3] == 0:
            return np.full(3, np.inf, np.float32)
        return self.basis_override.inverse().xform([-out[0], -out[1], out[2]])

    def update_camera_matrices(self) -> bool:
        """gaussian_splatting_rasterizer.gd:175-195: rebuild the 128-byte view+projection block if the camera
        moved; returns whether it changed."""
        xform = self.camera.get_camera_transform()
        fov, aspect, near, far = self.camera.get_camera_projection()
        key = (xform.tobytes(), self.basis_override.columns9().tobytes(), fov, aspect, near, far)
        if key == self._camera_key:
            return False
        lib = self._lib or _lib.load()
        out32 = np.zeros(32, np.float32)
        cam = np.zeros(3, np.float32)
        f32p = C.POINTER(C.c_float)
        bo = self.basis_override.columns9()
        _lib.check(lib.gsplat_make_view_proj(xform.ctypes.data_as(f32p), bo.ctypes.data_as(f32p), fov, aspect, near,
                                             far, out32.ctypes.data_as(f32p), cam.ctypes.data_as(f32p)),
                   "gsplat_make_view_proj")
        self._camera_key = key
        self.camera_push_constants = out32
        self._uniform_cam_pos = cam
        return True

    # -- debug info (main.gd:93-119) ----------------------------------------------------------------------
    def get_stats(self) -> dict:
        st = _lib.Stats()
        st.struct_size = C.sizeof(_lib.Stats)   # the library fills at most what this binding knows (gsplat.h)
        _lib.check(self._lib.gsplat_get_stats(self.context, C.byref(st)), "gsplat_get_stats")
        skip = ("algorithmic_bytes", "ms_kernel", "launches_kernel", "pairs_round", "struct_size", "reserved0")
        d = {name: getattr(st, name) for name, _ in _lib.Stats._fields_ if name not in skip}
        d["algorithmic_bytes"] = list(st.algorithmic_bytes)
        return d
