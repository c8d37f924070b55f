"""Synthetic scenes and cameras of SURVEY.md §8(d) / BASELINE.md §3.

Scenes are raw INRIA-style 62-float .ply rows (the layout util/ply_file.gd:41-69 consumes):
  [0:3] x y z | [3:6] normals | [6:9] f_dc | [9:54] f_rest (channel-major, 15 per channel) |
  [54] opacity logit | [55:58] log scale | [58:62] rotation quaternion (w, x, y, z), un-normalised.
"""
from dataclasses import dataclass

import numpy as np

ROW = 62

# name -> (N, sh_degree, width, height, seed); BASELINE.json configs (stand-ins where the scene is unavailable)
CONFIGS = {
    "c1": (100_000, 0, 1280, 720, 1),       # demo.ply stand-in
    "c2": (1_000_000, 0, 1920, 1080, 2),    # synthetic 1 M, SH deg 0
    "c3": (6_131_954, 3, 1920, 1080, 3),    # MipNeRF360 'bicycle'-like, SH deg 3
    "c4": (5_834_784, 3, 3840, 2160, 4),    # 'garden'-like, 4K
    "c5": (30_000_000, 0, 3840, 2160, 5),   # 30 M, 4K
    # c3 at the density of a real capture: same N / seed / camera, every splat 7.8x larger, so that a splat covers
    # ~9.4 tiles instead of 1.65 (D/N just under the reference's 10 N key budget, gaussian_splatting_rasterizer.gd:79)
    "c3d": (6_131_954, 3, 1920, 1080, 3),
    # ... and half way there: every splat 4x larger, D/N ~ 4.1 — part of the tiles saturate after a batch or two, most do
    # not: the regime between "one round wastes nothing" (c3) and "round A finishes every tile" (c3d)
    "c3m": (6_131_954, 3, 1920, 1080, 3),
}
SIZE_MULT = {"c3d": 7.8, "c3m": 4.0}  # splat-size multiplier on top of the SURVEY §8(d) law (1 for every BASELINE.json config)


def synthetic_rows(n: int, seed: int, sh_degree: int = 0, chunk=None, scale_n=None, size_mult=1.0) -> np.ndarray:
    """(n, 62) float32 rows.  RNG numpy default_rng(seed) (PCG64); draw order: pos, log-scale, quat,
    opacity logit, f_dc, f_rest.  `scale_n` fixes the splat-size law to a different N (used when a bounded
    sample of a large scene is generated); `size_mult` scales every splat (SIZE_MULT, "c3d")."""
    rng = np.random.default_rng(seed)
    ns = n if scale_n is None else scale_n
    s_lo = 0.002 * (1e6 / max(ns, 1)) ** (1.0 / 3.0) * size_mult
    rows = np.zeros((n, ROW), np.float32)
    rows[:, 0:3] = rng.normal(0.0, 1.5, (n, 3))
    rows[:, 55:58] = rng.uniform(np.log(s_lo), np.log(10.0 * s_lo), (n, 3))
    rows[:, 58:62] = rng.normal(0.0, 1.0, (n, 4))
    rows[:, 54] = rng.normal(0.0, 2.0, n)
    rows[:, 6:9] = rng.uniform(-1.5, 1.5, (n, 3))
    if sh_degree > 0:
        rest = rng.normal(0.0, 0.1, (n, 45)).astype(np.float32)
        k = (sh_degree + 1) ** 2 - 1  # coefficients per channel above band 0
        for ch in range(3):
            rows[:, 9 + 15 * ch: 9 + 15 * ch + k] = rest[:, 15 * ch: 15 * ch + k]
    return rows


def config_rows(name: str, n=None) -> np.ndarray:
    """The rows of a named configuration (the first n of them, drawn with the full scene's size law)."""
    full, deg, _, _, seed = CONFIGS[name]
    return synthetic_rows(full if n is None else n, seed, deg, scale_n=full, size_mult=SIZE_MULT.get(name, 1.0))


@dataclass
class CameraSpec:
    """Godot-world camera: camera-to-world basis columns X,Y,Z and origin O, perspective parameters."""
    basis: np.ndarray      # (3,3) columns = X,Y,Z
    origin: np.ndarray     # (3,)
    fov: float = 75.0      # Camera3D default fov (vertical, degrees)
    near: float = 0.05
    far: float = 4000.0

    def xform12(self) -> np.ndarray:
        return np.concatenate([self.basis[:, 0], self.basis[:, 1], self.basis[:, 2], self.origin]).astype(np.float32)


def default_camera(distance: float = 5.0) -> CameraSpec:
    """Godot camera at world (0,0,distance) looking down -Z at the origin."""
    return CameraSpec(basis=np.eye(3, dtype=np.float32), origin=np.array([0.0, 0.0, distance], np.float32))


def look_at_camera(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)) -> CameraSpec:
    """Godot Transform3D.looking_at: -Z points at the target."""
    eye = np.asarray(eye, np.float64)
    z = eye - np.asarray(target, np.float64)
    z /= np.linalg.norm(z)
    x = np.cross(np.asarray(up, np.float64), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return CameraSpec(basis=np.stack([x, y, z], axis=1).astype(np.float32), origin=eye.astype(np.float32))


def write_ply(path: str, rows: np.ndarray) -> None:
    """Write an INRIA-style binary little-endian .ply with the 62 float properties."""
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)]
             + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    rows = np.ascontiguousarray(rows, dtype="<f4").reshape(-1, ROW)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % rows.shape[0]
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rows.tobytes())
