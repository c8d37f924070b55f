"""godotgaussiansplatting_amd — MI355X-native forward Gaussian-splat rasterizer.

Drop-in for the compute-shader pipeline of 2Retr0/GodotGaussianSplatting
(util/gaussian_splatting_rasterizer.gd): the hot path is libgsplat_hip.so (hand-written HIP for
gfx950 behind the C ABI of include/gsplat.h); this package is the host-side mirror of the reference's
GDScript interface (GaussianSplattingRasterizer, PlyFile) plus the synthetic scenes of SURVEY.md §8(d).
There is no CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from .ply_file import PlyFile  # noqa: F401
from .godot_types import Camera3D, Texture2DRD, Basis  # noqa: F401
from .gaussian_splatting_rasterizer import GaussianSplattingRasterizer  # noqa: F401
from . import scenes  # noqa: F401
from .frame_ring import FrameRing  # noqa: F401

__all__ = ["PlyFile", "Camera3D", "Texture2DRD", "Basis", "GaussianSplattingRasterizer", "scenes", "FrameRing"]
