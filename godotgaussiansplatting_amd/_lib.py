"""ctypes binding of libgsplat_hip.so — exactly the declarations of include/gsplat.h.

The product path: no fallback.  `load()` raises if the shared library is missing (build it with
`python -m godotgaussiansplatting_amd.build` or `__graft_entry__.build()`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GSPLAT_LIB") or os.path.join(HERE, "libgsplat_hip.so")  # GSPLAT_LIB: A/B builds

GSPLAT_OK = 0
VERSION = (0 << 16) | 5   # GSPLAT_VERSION_MAJOR << 16 | GSPLAT_VERSION_MINOR of include/gsplat.h
FLAG_TIMING = 0x1
FLAG_FIX_LAST_TILE = 0x2
FLAG_FAST_EXP = 0x4
FLAG_KEEP_EMITTED = 0x8
FLAG_KERNEL_TIMING = 0x10
FLAG_BLOCK_CULL = 0x20
FLAG_TIES_STORAGE_ORDER = 0x40
FLAG_READBACK_RGB = 0x80
KERNEL_CLASSES = ['project', 'scan', 'emit', 'sort_upsweep', 'sort_spine', 'sort_downsweep', 'boundaries', 'render',
                  'splat_sort']
STRIPE_NONE, STRIPE_COLUMNS, STRIPE_ROWS = 0, 1, 2
NO_TARGET_TILE = 0xFFFFFFFF
(DEBUG_CULLED, DEBUG_KEYS_SORTED, DEBUG_VALUES_SORTED, DEBUG_TILE_BOUNDS, DEBUG_KEYS_EMITTED, DEBUG_VALUES_EMITTED,
 DEBUG_TILE_COUNTS, DEBUG_RECORDS, DEBUG_IMAGE, DEBUG_TILE_STAGED, DEBUG_BLOCK_SUMS, DEBUG_TILE_ORDER,
 DEBUG_SORT_RANK, DEBUG_EMIT_MODE, DEBUG_SLOT_IDS) = range(15)

# every symbol include/gsplat.h declares
EXPORTS = ["gsplat_create", "gsplat_create_view", "gsplat_destroy", "gsplat_upload_splats", "gsplat_upload_ply_rows",
           "gsplat_finalize_scene", "gsplat_resize",
           "gsplat_set_stripe", "gsplat_render", "gsplat_render_to", "gsplat_render_begin", "gsplat_render_end", "gsplat_pick", "gsplat_get_stats", "gsplat_set_timing", "gsplat_debug_read", "gsplat_debug_pow02",
           "gsplat_render_async", "gsplat_readback_wait", "gsplat_bind_external_image", "gsplat_export_image_fd",
           "gsplat_group_unique_id", "gsplat_group_create", "gsplat_group_create_local", "gsplat_group_set_cuts",
           "gsplat_group_render", "gsplat_group_exchanges_last_tile", "gsplat_group_destroy",
           "gsplat_create_batch_view", "gsplat_render_batch", "gsplat_render_batch_begin", "gsplat_render_batch_end",
           "gsplat_batch_image_device_ptr", "gsplat_group_render_batch",
           "gsplat_image_device_ptr", "gsplat_synchronize", "gsplat_make_view_proj", "gsplat_status_string",
           "gsplat_last_error", "gsplat_version"]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_splats", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("key_budget_factor", C.c_uint32), ("device_id", C.c_int32), ("flags", C.c_uint32),
                ("stripe_axis", C.c_uint32), ("stripe_begin", C.c_uint32), ("stripe_end", C.c_uint32),
                ("sh_degree", C.c_int32), ("stream", C.c_void_p)]


class Frame(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("cam_pos", C.c_float * 3),
                ("model_scale", C.c_float), ("time", C.c_float), ("heatmap_factor", C.c_float),
                ("target_tile", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("num_splats", C.c_uint64), ("num_visible", C.c_uint64), ("num_emitted", C.c_uint64),
                ("num_sorted", C.c_uint64), ("num_composited", C.c_uint64), ("capacity", C.c_uint64), ("overflow", C.c_int32),
                ("sort_passes", C.c_int32), ("sh_degree", C.c_int32), ("lazy_colors", C.c_int32),
                ("ms_projection", C.c_float), ("ms_sort", C.c_float), ("ms_boundaries", C.c_float),
                ("ms_render", C.c_float), ("ms_total", C.c_float), ("pair_key_bytes", C.c_int32),
                ("bytes_allocated", C.c_uint64),
                ("scene_bytes", C.c_uint64), ("algorithmic_bytes", C.c_uint64 * 4), ("ms_kernel", C.c_float * 9),
                ("launches_kernel", C.c_uint32 * 9), ("pairs_round", C.c_uint64 * 2),
                ("ms_gather", C.c_float), ("ms_readback", C.c_float)]


class GsplatError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME without
    the .7 of /opt/rocm's), so a process that uses both this library and torch (the multi-GPU host:
    torch.distributed/RCCL) would otherwise initialise two runtimes and the second one sees no GPU.  If torch is
    installed, its bundled runtime is loaded first with RTLD_GLOBAL: libgsplat_hip.so's HIP symbols then bind to
    it (global scope is searched before the library's own DT_NEEDED), whichever of the two is imported first.
    GSPLAT_HIP_RUNTIME=system skips this (standalone use against /opt/rocm)."""
    if os.environ.get("GSPLAT_HIP_RUNTIME", "torch") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def share_rccl_with_torch():
    """One RCCL per process, like the HIP runtime above: gsplat_group_* load librccl on first use and take a copy that
    is already in the process.  With torch installed that should be torch's own (built against the runtime the process
    runs on), so it is loaded here, globally, before the first group call."""
    if os.environ.get("GSPLAT_HIP_RUNTIME", "torch") == "system" or os.environ.get("GSPLAT_RCCL_LIB"):
        return
    try:
        import importlib.util
        import sys
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        if "torch" not in sys.modules:
            # torch first: a process that loads torch's librccl by hand and imports torch AFTERWARDS ends with a double
            # free in the libraries' exit handlers (seen on ROCm 7.2 / torch 2.10: pytest -k group, round 4)
            import torch  # noqa: F401
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            os.environ["GSPLAT_RCCL_LIB"] = cand
    except Exception:
        pass


def load():
    """Load libgsplat_hip.so and declare its prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing — build it first (python -m godotgaussiansplatting_amd.build); "
                           "there is no CPU fallback for the hot path")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(SO_PATH)
    f32p, vp, u32 = C.POINTER(C.c_float), C.c_void_p, C.c_uint32
    lib.gsplat_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.gsplat_create_view.argtypes = [vp, C.POINTER(Config), C.POINTER(vp)]
    lib.gsplat_destroy.argtypes = [vp]
    lib.gsplat_upload_splats.argtypes = [vp, u32, u32, vp]
    lib.gsplat_upload_ply_rows.argtypes = [vp, u32, u32, vp, C.c_float]
    lib.gsplat_finalize_scene.argtypes = [vp]
    lib.gsplat_resize.argtypes = [vp, u32, u32]
    lib.gsplat_set_stripe.argtypes = [vp, u32, u32, u32]
    lib.gsplat_render.argtypes = [vp, C.POINTER(Frame), vp]
    lib.gsplat_render_to.argtypes = [vp, C.POINTER(Frame), vp, u32, u32, u32]
    lib.gsplat_render_begin.argtypes = [vp, C.POINTER(Frame), vp]
    lib.gsplat_render_end.argtypes = [vp, vp, u32, u32, u32, vp]
    lib.gsplat_pick.argtypes = [vp, C.POINTER(Frame), u32, f32p]
    lib.gsplat_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.gsplat_set_timing.argtypes = [vp, u32]
    lib.gsplat_debug_read.argtypes = [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gsplat_debug_pow02.argtypes = [vp, u32, C.c_uint64, f32p]
    lib.gsplat_render_async.argtypes = [vp, C.POINTER(Frame), C.POINTER(C.c_uint64)]
    lib.gsplat_readback_wait.argtypes = [vp, C.c_uint64, C.POINTER(f32p)]
    lib.gsplat_bind_external_image.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64]
    lib.gsplat_export_image_fd.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    lib.gsplat_group_unique_id.argtypes = [vp]
    lib.gsplat_group_create.argtypes = [vp, vp, C.c_int, C.c_int, u32, C.POINTER(vp)]
    lib.gsplat_group_create_local.argtypes = [C.POINTER(vp), C.c_int, u32, C.POINTER(vp)]
    lib.gsplat_group_set_cuts.argtypes = [vp, C.POINTER(u32)]
    lib.gsplat_group_render.argtypes = [vp, C.POINTER(Frame), C.POINTER(vp)]
    lib.gsplat_group_exchanges_last_tile.argtypes = [vp]
    lib.gsplat_group_destroy.argtypes = [vp]
    lib.gsplat_create_batch_view.argtypes = [vp, C.POINTER(Config), u32, C.POINTER(vp)]
    lib.gsplat_render_batch.argtypes = [vp, C.POINTER(Frame), u32]
    lib.gsplat_render_batch_begin.argtypes = [vp, C.POINTER(Frame), u32, vp]
    lib.gsplat_render_batch_end.argtypes = [vp, vp]
    lib.gsplat_batch_image_device_ptr.argtypes = [vp, u32, C.POINTER(vp)]
    lib.gsplat_group_render_batch.argtypes = [vp, C.POINTER(Frame), u32]
    lib.gsplat_image_device_ptr.argtypes = [vp, C.POINTER(vp)]
    lib.gsplat_synchronize.argtypes = [vp]
    lib.gsplat_make_view_proj.argtypes = [f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float, f32p, f32p]
    lib.gsplat_status_string.restype = C.c_char_p
    lib.gsplat_status_string.argtypes = [C.c_int]
    lib.gsplat_last_error.restype = C.c_char_p
    lib.gsplat_version.restype = C.c_uint32
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("gsplat_status_string", "gsplat_last_error", "gsplat_version"):
            fn.restype = C.c_int
    # the mirrors above are those of header version 0.5: refuse a library that was built from another one (a stale
    # GSPLAT_LIB) instead of reading shifted fields
    have = lib.gsplat_version()
    if have != VERSION:
        raise RuntimeError(f"{SO_PATH} is version {have >> 16}.{have & 0xFFFF}, this binding is for "
                           f"{VERSION >> 16}.{VERSION & 0xFFFF}")
    _lib = lib
    return lib


def check_value(status, where):
    """For entry points that return a non-negative value on success."""
    if status < 0:
        check(status, where)
    return status


def check(status, where):
    if status != GSPLAT_OK:
        lib = load()
        msg = lib.gsplat_status_string(status).decode()
        detail = lib.gsplat_last_error().decode()
        raise GsplatError(status, where, f"{msg}{'; ' + detail if detail else ''}")
