"""PlyFile — host-side mirror of util/ply_file.gd.

Same surface: `PlyFile(path)` / `parse`, `size`, `vertices`, `properties`, `get_vertex`, and the static
`load_gaussian_splats(point_cloud, stride, device, buffer, should_terminate, num_loaded, callback)` that
uploads the scene in `stride`-sized chunks from worker threads while frames may already be rendering.
Unlike the reference, the per-vertex swizzle (ply_file.gd:41-69) is NOT done on the CPU: raw 62-float
rows go to the GPU and gsplat_upload_ply_rows converts them there.
"""
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .scenes import ROW


class PlyFile:
    def __init__(self, path: str = ""):
        self.size = 0
        self.vertices = np.zeros(0, np.float32)
        self.properties = []
        self.big_endian = False
        if path:
            self.parse(path)

    @classmethod
    def from_rows(cls, rows: np.ndarray) -> "PlyFile":
        """In-memory scene (synthetic generators)."""
        p = cls()
        r = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, ROW)
        p.size = r.shape[0]
        p.vertices = r.reshape(-1)
        p.properties = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
                        + [f"f_rest_{i}" for i in range(45)] + ["opacity"] + [f"scale_{i}" for i in range(3)]
                        + [f"rot_{i}" for i in range(4)])
        return p

    def parse(self, path: str) -> None:
        """ply_file.gd:10-19: naive header walk — `format`, `element`, `property` lines until end_header,
        then size*len(properties) float32 values."""
        with open(path, "rb") as f:
            line = f.readline().decode("ascii", "replace").strip().split(" ")
            while line[0] != "end_header":
                raw = f.readline()
                if not raw:
                    raise ValueError("ply: end_header not found")
                line = raw.decode("ascii", "replace").strip().split(" ")
                if line[0] == "format":
                    self.big_endian = line[1] == "binary_big_endian"
                elif line[0] == "element":
                    self.size = int(line[2])
                elif line[0] == "property":
                    self.properties.append(line[2])
            count = self.size * len(self.properties)
            data = np.frombuffer(f.read(count * 4), dtype=">f4" if self.big_endian else "<f4")
        if data.size != count:
            raise ValueError(f"ply: expected {count} floats, file holds {data.size}")
        self.vertices = data.astype(np.float32)

    def get_vertex(self, index: int) -> dict:
        n = len(self.properties)
        start = n * index
        return {self.properties[i]: float(self.vertices[start + i]) for i in range(n)}

    def rows(self) -> np.ndarray:
        """(size, 62) view.  Like the reference the 62-property INRIA order is assumed."""
        if len(self.properties) != ROW:
            raise ValueError(f"ply: {len(self.properties)} properties per vertex, the loader needs the 62 INRIA ones")
        return self.vertices.reshape(self.size, ROW)

    @staticmethod
    def load_gaussian_splats(point_cloud: "PlyFile", stride: int, device, buffer, should_terminate_reference: list,
                             num_points_loaded: list, callback, time_source=None, max_workers: int = 4) -> None:
        """ply_file.gd:28-77.  `device.buffer_update_ply_rows(buffer, first, rows, creation_time)` plays the
        role of RenderingDevice.buffer_update.  The reference divides by `stride` (= size/1000) and crashes
        for scenes under 1000 splats; here stride is clamped to >= 1."""
        assert len(should_terminate_reference) == 1 and len(num_points_loaded) == 1
        stride = max(1, int(stride))
        rows = point_cloud.rows()
        mutex = threading.Lock()
        now = time_source if time_source is not None else (lambda: time.monotonic())

        def task(i: int):
            if should_terminate_reference[0]:
                return
            first = i * stride
            tile_size = min(point_cloud.size - first, stride)
            if tile_size <= 0:
                return
            creation_time = float(now())
            if should_terminate_reference[0]:
                return
            device.buffer_update_ply_rows(buffer, first, rows[first:first + tile_size], creation_time)
            with mutex:
                num_points_loaded[0] += tile_size

        num_tasks = -(-point_cloud.size // stride)
        with ThreadPoolExecutor(max_workers=max_workers) as pool:
            list(pool.map(task, range(num_tasks)))
        callback()
