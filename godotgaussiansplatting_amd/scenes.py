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
# c3r: c3's N, band count, frame and camera with the STATISTICS of a trained capture instead of a Gaussian blob of
# near-isotropic splats (capture_like_rows): surfaces, flat anisotropic splats, a log-normal size law with a heavy tail,
# bimodal opacity, far-field floaters — D/N ~ 8 (the reference budgets 10 N keys, gaussian_splatting_rasterizer.gd:79),
# long-tailed pairs per tile.  Still a stand-in: the INRIA captures BASELINE.json names cannot be fetched.
CONFIGS["c3r"] = (6_131_954, 3, 1920, 1080, 31)
CAPTURE_LIKE = {"c3r"}


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


def _quat_z_to(normals: np.ndarray) -> np.ndarray:
    """Unit quaternions (w, x, y, z) rotating +z onto each unit normal (half-way construction; antipodal: about x)."""
    n = normals / np.maximum(np.linalg.norm(normals, axis=1, keepdims=True), 1e-12)
    w = 1.0 + n[:, 2]
    q = np.stack([w, -n[:, 1], n[:, 0], np.zeros_like(w)], axis=1)
    flip = w < 1e-6
    q[flip] = np.array([0.0, 1.0, 0.0, 0.0])
    return q / np.linalg.norm(q, axis=1, keepdims=True)


CAPTURE_SIZE = 0.0125  # median tangential splat size of capture_like_rows, scene units: calibrated to D/N ~ 8 at 1080p from
                       # the default camera (tools/calibrate_c3r.py) — absolute, not a function of N: a bounded sample of
                       # the scene (bench.py cpu_baseline) is a thinner version of the same surfaces


def capture_like_rows(n: int, seed: int, sh_degree: int = 3, size=CAPTURE_SIZE) -> np.ndarray:
    """(n, 62) float32 rows with the statistics of a trained 3DGS capture (MipNeRF360-like: an object on a ground plane in
    a room-sized background), every draw from default_rng(seed) in a fixed order.  Population (interleaved in id order, as a
    trained model's densification leaves it): 45 % ground disk, 25 % shells of three objects, 10 % a back wall, 12 %
    volumetric clutter, 8 % far-field floaters.  Splats on a surface are FLAT discs in its tangent plane (normal axis a
    tenth of the tangential ones, which differ by up to 3:1 among themselves -> anisotropy up to ~10:1 and beyond);
    sizes log-normal (sigma 0.7) with a 2 % heavy tail of 4-20x (background blobs); sizes grow with the distance from
    the scene's centre as a trained model's do; opacity bimodal (half near-opaque, half faint)."""
    rng = np.random.default_rng(seed)
    rows = np.zeros((n, ROW), np.float32)
    kind = rng.choice(5, size=n, p=[0.45, 0.25, 0.10, 0.12, 0.08])
    pos = np.zeros((n, 3))
    nrm = rng.normal(0.0, 1.0, (n, 3))                      # clutter / floaters: random orientation
    u = rng.uniform(0.0, 1.0, (n, 3))
    g = rng.normal(0.0, 1.0, (n, 3))
    # ground: a disk of radius 4.5 at y = -1.2, denser towards the centre, 1 cm of roughness
    m = kind == 0
    r, a = 4.5 * u[m, 0] ** 0.7, 2.0 * np.pi * u[m, 1]
    pos[m] = np.stack([r * np.cos(a), -1.2 + 0.01 * g[m, 0], r * np.sin(a)], axis=1)
    nrm[m] = np.array([0.0, 1.0, 0.0]) + 0.05 * g[m]
    # three objects: spherical shells (radius 0.9, 0.5, 0.35) standing on / near the ground
    m = kind == 1
    which = (u[m, 2] * 3).astype(int).clip(0, 2)
    centre = np.array([[0.0, -0.3, 0.0], [1.4, -0.7, 0.6], [-1.2, -0.85, -0.4]])[which]
    radius = np.array([0.9, 0.5, 0.35])[which]
    d = g[m] / np.maximum(np.linalg.norm(g[m], axis=1, keepdims=True), 1e-9)
    pos[m] = centre + d * (radius * (1.0 + 0.01 * rng.normal(0.0, 1.0, m.sum())))[:, None]
    nrm[m] = d
    # back wall: the plane z = -4, 9 x 4 units
    m = kind == 2
    pos[m] = np.stack([9.0 * (u[m, 0] - 0.5), -1.2 + 4.0 * u[m, 1], -4.0 + 0.01 * g[m, 0]], axis=1)
    nrm[m] = np.array([0.0, 0.0, 1.0]) + 0.05 * g[m]
    # clutter: a blob around the objects
    m = kind == 3
    pos[m] = g[m] * np.array([1.6, 0.7, 1.6]) + np.array([0.0, -0.4, 0.0])
    # floaters: a shell of radius 6..30 all around (some behind the camera, some in front of everything)
    m = kind == 4
    d = g[m] / np.maximum(np.linalg.norm(g[m], axis=1, keepdims=True), 1e-9)
    pos[m] = d * (6.0 + 24.0 * u[m, 0] ** 2)[:, None]
    # (none within 3 units of where the cameras stand — 5 units from the centre: training views see through that space)
    too_close = m & (np.linalg.norm(pos, axis=1) < 8.5) & (np.abs(np.linalg.norm(pos, axis=1) - 5.0) < 3.0)
    pos[too_close] *= (8.5 / np.maximum(np.linalg.norm(pos[too_close], axis=1), 1e-9))[:, None]
    rows[:, 0:3] = pos
    # sizes: log-normal, heavy tail, growing with distance from the centre (floaters are big and faint)
    base = size * np.exp(0.7 * rng.normal(0.0, 1.0, n))
    tail = rng.uniform(0.0, 1.0, n) < 0.02
    base = np.where(tail, base * rng.uniform(4.0, 20.0, n), base)
    base *= np.maximum(1.0, np.linalg.norm(pos, axis=1) / 3.0)
    aniso = np.exp(rng.uniform(-0.55, 0.55, (n, 2)))        # tangential axes: up to 3:1 between them
    flat = np.where(kind <= 2, 0.1, np.exp(rng.uniform(-0.5, 0.5, n)))   # surface splats: normal axis a tenth
    rows[:, 55] = np.log(base * aniso[:, 0])
    rows[:, 56] = np.log(base * aniso[:, 1])
    rows[:, 57] = np.log(base * flat)
    rows[:, 58:62] = _quat_z_to(nrm) * rng.uniform(0.5, 2.0, (n, 1))   # (un-normalised, as in a trained file)
    # opacity: bimodal; floaters faint
    hi = rng.uniform(0.0, 1.0, n) < 0.5
    rows[:, 54] = np.where(hi, rng.normal(3.0, 1.0, n), rng.normal(-2.0, 1.0, n))
    rows[kind == 4, 54] -= 1.0
    rows[:, 6:9] = rng.uniform(-1.5, 1.5, (n, 3))
    rest = rng.normal(0.0, 0.1, (n, 45)).astype(np.float32)
    if sh_degree > 0:
        k = (sh_degree + 1) ** 2 - 1
        for ch in range(3):
            rows[:, 9 + 15 * ch: 9 + 15 * ch + k] = rest[:, 15 * ch: 15 * ch + k]
    return rows


def config_rows(name: str, n=None) -> np.ndarray:
    """The rows of a named configuration (the first n of them, drawn with the full scene's size law)."""
    full, deg, _, _, seed = CONFIGS[name]
    if name in CAPTURE_LIKE:
        # (a sample of a capture-like scene is the same surfaces sampled more thinly — not a prefix of the full draw)
        return capture_like_rows(full if n is None else n, seed, deg)
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
