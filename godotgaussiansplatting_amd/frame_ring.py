"""FrameRing — keep several frames in flight on one GPU.

The reference renders through Godot's RenderingDevice, which keeps (by default) two frames queued on the GPU; with
HIP the same is expressed as a ring of contexts on ONE scene (gsplat_create_view): each owns a stream and its own
intermediate buffers (RasterizeData, the sort buffers, tile ranges, image), all read the same splat buffer, and
consecutive frames go to consecutive contexts.  Nothing is skipped or shared between frames — every frame runs the
whole pipeline — but the HBM-bound passes of frame k+1 can overlap the issue-bound compositing of frame k
(DESIGN.md §7).  Latency of a single frame is unchanged; the scene is uploaded and stored once.
"""
from . import capi


class FrameRing:
    def __init__(self, depth, max_splats, width, height, **ctx_kwargs):
        first = capi.Context(max_splats, width, height, **ctx_kwargs)
        ctx_kwargs.pop("device_id", None)
        self.contexts = [first] + [first.view(**ctx_kwargs) for _ in range(max(1, int(depth)) - 1)]
        self._turn = 0

    def upload_ply_rows(self, rows, first=0, load_time=-10.0):
        self.contexts[0].upload_ply_rows(rows, first=first, load_time=load_time)  # one scene behind every slot

    def upload_splats(self, records, first=0):
        self.contexts[0].upload_splats(records, first=first)

    def finalize_scene(self):
        self.contexts[0].finalize_scene()

    def render(self, frame, out=None):
        """Enqueue one frame on the next context of the ring; returns that context (its image / taps hold the frame
        once it has been synchronised)."""
        c = self.contexts[self._turn % len(self.contexts)]
        self._turn += 1
        c.render(frame, out)
        return c

    def synchronize(self):
        for c in self.contexts:
            c.synchronize()

    def close(self):
        for c in reversed(self.contexts):
            c.close()
        self.contexts = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
