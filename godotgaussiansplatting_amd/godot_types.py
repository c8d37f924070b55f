"""Minimal stand-ins for the Godot objects the reference's rasterizer class receives
(Camera3D, Texture2DRD, Basis) — just the members util/gaussian_splatting_rasterizer.gd touches."""
import numpy as np


class Basis:
    """3x3 basis stored as columns (Godot's Basis.x/.y/.z)."""
    IDENTITY = None

    def __init__(self, cols=None):
        self.m = np.eye(3, dtype=np.float32) if cols is None else np.asarray(cols, np.float32).reshape(3, 3)

    def columns9(self):
        return np.concatenate([self.m[:, 0], self.m[:, 1], self.m[:, 2]]).astype(np.float32)

    def inverse(self):
        return Basis(np.linalg.inv(self.m.astype(np.float64)).astype(np.float32))

    def xform(self, v):
        return (self.m.astype(np.float64) @ np.asarray(v, np.float64)).astype(np.float32)

    def is_identity(self):
        return np.array_equal(self.m, np.eye(3, dtype=np.float32))


Basis.IDENTITY = Basis()


class Camera3D:
    """global transform + perspective parameters (Godot 4.3 defaults: fov 75, near 0.05, far 4000)."""

    def __init__(self, basis=None, origin=(0.0, 0.0, 0.0), fov=75.0, near=0.05, far=4000.0, aspect=16.0 / 9.0):
        self.basis = np.eye(3, dtype=np.float32) if basis is None else np.asarray(basis, np.float32).reshape(3, 3)
        self.global_position = np.asarray(origin, np.float32)
        self.fov, self.near, self.far, self.aspect = float(fov), float(near), float(far), float(aspect)

    @classmethod
    def from_spec(cls, spec, aspect):
        return cls(spec.basis, spec.origin, spec.fov, spec.near, spec.far, aspect)

    def get_camera_transform(self):
        """12 floats: basis columns X, Y, Z then origin."""
        return np.concatenate([self.basis[:, 0], self.basis[:, 1], self.basis[:, 2],
                               self.global_position]).astype(np.float32)

    def get_camera_projection(self):
        """Perspective parameters (fovy degrees, aspect, near, far); the matrix itself is built by
        gsplat_make_view_proj exactly as Godot's Projection::set_perspective does."""
        return (self.fov, self.aspect, self.near, self.far)


class Texture2DRD:
    """The output texture handle of the reference (its RID is set to an RGBA32F storage texture,
    gaussian_splatting_rasterizer.gd:92,101).  Here: a device pointer + size, and a host read-back."""

    def __init__(self):
        self.texture_rd_rid = 0   # device pointer of the RGBA32F image (0 = invalid RID)
        self.size = (0, 0)
        self._reader = None

    def get_image(self) -> np.ndarray:
        """(H, W, 4) float32 copy of the rendered frame."""
        if not self._reader:
            raise RuntimeError("texture has no backing image (rasterizer not initialised)")
        return self._reader()
