"""Tile-stripe sharding of one frame across the GPUs of a node (no reference counterpart —
the reference is single-GPU).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The scene is replicated; rank r
owns a contiguous stripe of tile columns (or rows).  Its context clamps every splat's tile rectangle to the
stripe, so the per-tile key sets — and therefore the pixels — are exactly those of the single-GPU frame.
The only exchange step is the gather of the finished tiles: every rank renders its stripe (gsplat_render_to), packs
the three colour channels — alpha is the constant 1.0 of gsplat_render.glsl:101, so 12 of the 16 bytes per pixel
travel — into its slot of a stripe-major staging tensor, and one all_gather_into_tensor fills the others (stripes
are padded to the widest one so the collective is a plain equal-size all-gather: with 7 direct xGMI links per GPU
each peer's slot arrives on its own link).  `unstripe` then assembles the row-major frame.

The partition / gather logic is backend-agnostic (tests drive it with gloo on CPU and a stand-in renderer).
"""
from dataclasses import dataclass

import numpy as np

TILE = 16


def even_cuts(num_tiles: int, world: int):
    """world+1 stripe boundaries, tile units, as even as possible."""
    return [(i * num_tiles) // world for i in range(world + 1)]


def balanced_cuts(weights, world: int, min_width: int = 1):
    """Boundaries that equalise the summed per-column (or per-row) weight — e.g. last frame's pairs per tile
    column plus a constant per tile.  Every stripe keeps at least `min_width` tiles when possible."""
    w = np.asarray(weights, np.float64)
    n = w.size
    if n < world * min_width:
        return even_cuts(n, world)
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(csum, target, side="left"))
        # pick the nearer of c-1 / c
        if c > 0 and abs(csum[c - 1] - target) <= abs(csum[min(c, n)] - target):
            c -= 1
        c = max(c, cuts[-1] + min_width)
        c = min(c, n - (world - r) * min_width)
        cuts.append(c)
    cuts.append(n)
    return cuts


def time_balanced_cuts(cuts, times, prior=None, min_width: int = 1):
    """Boundaries from the MEASURED frame time of every rank under the current ones.  A rank's time has parts no static
    profile sees — the splats it has to project for a WIDE sparse stripe at the frame's edge (c4, 8 stripes balanced by
    pairs: the outer ranks project twice what the middle ones do and take 0.38 ms against 0.29), the serial depth of its
    heaviest tile, its fixed launches — so the stripes are re-cut from what the ranks actually took: inside rank r's stripe
    the cost per tile row (or column) is taken as times[r] / width_r, shaped by `prior` (per-unit weights, e.g. pairs per
    tile row + a constant; None: flat), and the new boundaries equalise its integral.  Iterating converges to equal times
    (a fixed point: equal times reproduce the same cuts).  Every rank computes the same cuts from the same gathered
    times.  A rank without tiles keeps zero cost for its (empty) range."""
    cuts = [int(c) for c in cuts]
    world = len(cuts) - 1
    n = cuts[-1]
    shape = np.ones(n, np.float64) if prior is None else np.maximum(np.asarray(prior, np.float64), 1e-12)
    cost = np.zeros(n, np.float64)
    for r in range(world):
        a, b = cuts[r], cuts[r + 1]
        if b > a:
            seg = shape[a:b]
            cost[a:b] = float(times[r]) * seg / seg.sum()
    return balanced_cuts(cost, world, min_width)


class BatchSubmitter:
    """Frames handed over one at a time, submitted `batch` at a time (gsplat_group_render_batch: B consecutive frames of a rank
    through ONE launch sequence, one all-reduce of B words, one all-gather-v of B stripes), round-robin over the `groups` —
    one per launch sequence in flight, each with its own communicator.  `flush()` submits what is left as a partial batch,
    so that a run of K frames renders exactly K.  batch == 1: gsplat_group_render, frame by frame.
    `groups`: objects with render(frame) and render_batch(list of frames) (capi.Group)."""

    def __init__(self, groups, batch=1):
        self.groups = list(groups)
        self.batch = max(1, int(batch))
        self.pending = []
        self.turn = 0
        self.frames_submitted = 0
        self.submissions = 0

    def _submit(self):
        g = self.groups[self.turn % len(self.groups)]
        if self.batch > 1:
            g.render_batch(list(self.pending))
        else:
            g.render(self.pending[0])
        self.frames_submitted += len(self.pending)
        self.submissions += 1
        self.pending.clear()
        self.turn += 1

    def step(self, frame):
        self.pending.append(frame)
        if len(self.pending) >= self.batch:
            self._submit()

    def flush(self):
        if self.pending:
            self._submit()


@dataclass
class StripeLayout:
    """Geometry of the stripe-major staging buffer."""
    axis: str            # "columns" | "rows"
    width: int
    height: int
    cuts: list           # world+1 boundaries in tiles

    @property
    def world(self):
        return len(self.cuts) - 1

    def px_range(self, r):
        lim = self.width if self.axis == "columns" else self.height
        return min(self.cuts[r] * TILE, lim), min(self.cuts[r + 1] * TILE, lim)

    @property
    def slot_px(self):
        """Padded stripe extent (pixels along the split axis) = the widest stripe."""
        return max(1, max(self.px_range(r)[1] - self.px_range(r)[0] for r in range(self.world)))

    def slot_shape(self, channels=4):
        """Shape of one rank's slot in the staging tensor (rows, cols, channels)."""
        return (self.height, self.slot_px, channels) if self.axis == "columns" else (self.slot_px, self.width, channels)

    def slot_pitch_px(self):
        return self.slot_shape()[1]

    def slot_origin(self, r):
        a, _ = self.px_range(r)
        return (a, 0) if self.axis == "columns" else (0, a)


def unstripe(staging, layout: StripeLayout, out):
    """staging: (world, *slot_shape) -> out: (H, W, 4) row-major.  Works on torch tensors and NumPy arrays.  A
    3-channel staging (RGB-only gather) fills the colour channels; out's alpha plane is left as it is (pre-set to 1)."""
    ch = staging.shape[-1]
    for r in range(layout.world):
        a, b = layout.px_range(r)
        if b <= a:
            continue
        if layout.axis == "columns":
            out[:, a:b, :ch] = staging[r, :, : b - a, :]
        else:
            out[a:b, :, :ch] = staging[r, : b - a, :, :]
    return out


def shared_torch_stream():
    """Make a NON-default torch stream current and return (torch stream, raw hipStream_t handle) for
    capi.Context(stream=...).  torch's default stream has handle 0, which the C ABI reads as "create my own stream";
    a context that is to run in torch's stream order (no host sync before RCCL) therefore needs a real stream."""
    import torch
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    return s, s.cuda_stream


class StripeRasterizer:
    """One rank's share of the frame.  `ctx` is a capi.Context holding the whole (replicated) scene — or a LIST of such
    contexts, each created on its own torch stream (`streams`), to keep several frames in flight on the rank
    (render_pipelined alternates between them; per-rank work at 4-8 GPUs is small and latency-bound, so overlapping
    consecutive frames pays as it does on one GPU with FrameRing).
    The rendering and the per-tile counts go through two small hooks (`_render_stripe`, `_tile_counts`) so the
    partition/gather logic can be driven on CPU with gloo and a stand-in renderer (tests)."""

    def __init__(self, ctx, width, height, rank, world, axis="columns", group=None, device=None,
                 sync_after_render=True, host_staged_gather=False, streams=None, exchange_last_tile=False,
                 rgb_only=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctxs = list(ctx) if isinstance(ctx, (list, tuple)) else [ctx]
        self.ctx = self.ctxs[0]
        self.streams = list(streams) if streams is not None else [None] * len(self.ctxs)
        assert len(self.streams) == len(self.ctxs)
        self.rank, self.world, self.group = rank, world, group
        # sync_after_render=False when every context was created on its torch stream (stream order suffices);
        # host_staged_gather=True gathers through host memory (for backends without device collectives, e.g. a
        # gloo functional test of several ranks on one GPU) — never the benchmarked path
        self.sync_after_render, self.host_staged_gather = sync_after_render, host_staged_gather
        # exchange_last_tile=True for contexts created with FLAG_BLOCK_CULL: a rank that skips whole blocks of the
        # scene only knows its own stripe's highest populated tile, and quirk Q5/Q6 needs the frame's — the frame is
        # then rendered as gsplat_render_begin / 4-byte all-reduce(MAX) / gsplat_render_end
        self.exchange_last_tile = bool(exchange_last_tile)
        # rgb_only: ship 12 B per pixel (alpha == 1.0 everywhere, gsplat_render.glsl:101): -25 % bytes on the links
        self.channels = 3 if rgb_only else 4
        # its own communicator: collectives of one communicator run in issue order on one stream, so on the frame
        # group the 4-byte all-reduce of frame k+1 would queue behind the all-gather of frame k (which waits for
        # frame k's compositor) and serialise the frames in flight
        self.exchange_group = None
        if self.exchange_last_tile and world > 1:
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.exchange_group = dist.new_group(ranks=ranks)
        self.width, self.height, self.axis = width, height, axis
        self.gx, self.gy = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.depth = max(2, len(self.ctxs))  # staging slots: >= 2 so a gather can overlap the next render
        self.frame_outs = [torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
                           for _ in range(self.depth)]
        for f in self.frame_outs:
            f[..., 3] = 1.0  # gsplat_render.glsl:101: alpha is 1.0 in every pixel, rendered or not
        self.frame_out = self.frame_outs[0]
        self.last_tile = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(self.depth)]
        self.set_cuts(even_cuts(self.gx if axis == "columns" else self.gy, world))

    # ---- hooks ---------------------------------------------------------------------------------------
    def _apply_stripe(self, begin, end):
        from . import capi
        ax = capi.STRIPE_COLUMNS if self.axis == "columns" else capi.STRIPE_ROWS
        for c in self.ctxs:
            c.set_stripe(ax, begin, end)

    def _render_stripe(self, frame, slot, ctx=None):
        """Render this rank's tiles into `slot` (a contiguous (rows, cols, 4) tensor)."""
        ctx = ctx if ctx is not None else self.ctx
        ox, oy = self.layout.slot_origin(self.rank)
        ctx.render_to(frame, slot.data_ptr(), self.layout.slot_pitch_px(), ox, oy)
        if self.sync_after_render:
            ctx.synchronize()  # the context renders on its own stream; RCCL runs on torch's

    def _render_begin(self, frame, ctx, word):
        """First half of the stripe's frame; the stripe's own 'highest populated tile + 1' lands in `word`."""
        ctx.render_begin(frame, word.data_ptr())
        if self.sync_after_render:
            ctx.synchronize()

    def _render_end(self, slot, ctx, word):
        """Second half, with the frame-global value in `word`."""
        if self.sync_after_render:
            self.torch.cuda.current_stream().synchronize()  # the all-reduce ran on torch's stream
        ox, oy = self.layout.slot_origin(self.rank)
        ctx.render_end(slot.data_ptr(), self.layout.slot_pitch_px(), ox, oy, word.data_ptr())
        if self.sync_after_render:
            ctx.synchronize()

    def _max_over_ranks(self, word):
        if self.host_staged_gather:
            h = word.detach().to("cpu")
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.exchange_group or self.group)
            word.copy_(h)
        else:
            self.dist.all_reduce(word, op=self.dist.ReduceOp.MAX, group=self.exchange_group or self.group)

    def _render_rank(self, frame, slot, ctx, k):
        """This rank's part of frame k: its stripe (if it has one) and, with exchange_last_tile, its part in the
        4-byte all-reduce — a collective, so ranks without tiles take part too."""
        a, b = self.layout.px_range(self.rank)
        if not self.exchange_last_tile:
            if b > a:
                self._render_stripe(frame, slot, ctx)
            return
        word = self.last_tile[k]
        if b > a:
            self._render_begin(frame, ctx, word)
        else:
            word.zero_()
        self._max_over_ranks(word)
        if b > a:
            self._render_end(slot, ctx, word)

    def _tile_counts(self):
        b = self._last_ctx.read_bounds().astype(np.int64)
        return np.clip(b[:, 1] - b[:, 0], 0, None).reshape(self.gy, self.gx)

    def _local_bounds(self):
        """(this rank's tile_bounds tap, its pair count) of the last frame."""
        return self._last_ctx.read_bounds(), int(self._last_ctx.stats()["num_sorted"])

    def _on_stream(self, k):
        import contextlib
        s = self.streams[k % len(self.streams)]
        return self.torch.cuda.stream(s) if s is not None else contextlib.nullcontext()

    # ---- partition -----------------------------------------------------------------------------------
    def set_cuts(self, cuts):
        torch = self.torch
        self.flush_all()
        self.layout = StripeLayout(self.axis, self.width, self.height, list(cuts))
        self._apply_stripe(cuts[self.rank], cuts[self.rank + 1])
        shape = (self.world,) + self.layout.slot_shape(self.channels)
        self.staging = [torch.zeros(shape, dtype=torch.float32, device=self.device) for _ in range(self.depth)]
        self.packed = [torch.zeros(self.layout.slot_shape(self.channels), dtype=torch.float32, device=self.device)
                       for _ in range(self.depth)] if self.channels != 4 else None
        self.slot = [torch.zeros(self.layout.slot_shape(), dtype=torch.float32, device=self.device)
                     for _ in range(self.depth)]
        if any(s is not None for s in self.streams):
            torch.cuda.synchronize()  # the new buffers were zeroed on the current stream; the ring streams use them
        self._turn = 0
        self._pending = []
        self._last_ctx = self.ctx

    def _gather(self, st, slot, async_op, k=0):
        if self.packed is not None:  # RGBA32F stripe -> its three colour planes, contiguous (a few MB, one copy kernel)
            self.packed[k].copy_(slot[..., : self.channels])
            slot = self.packed[k]
        if self.host_staged_gather:
            host_in = slot.detach().to("cpu").contiguous()
            host_out = self.torch.empty((self.world,) + tuple(slot.shape), dtype=slot.dtype)
            self.dist.all_gather_into_tensor(host_out.view(-1), host_in.view(-1), group=self.group)
            st.copy_(host_out)
            return None
        return self.dist.all_gather_into_tensor(st.view(-1), slot.view(-1), group=self.group, async_op=async_op)

    def render(self, frame, assemble=True, async_gather=False):
        """Render this rank's stripe and all-gather the frame.  Returns the (H,W,4) device tensor (every rank
        ends up with the full frame, like the reference's single render texture)."""
        self.flush_all()
        k = self._turn % self.depth
        self._turn += 1
        st, slot = self.staging[k], self.slot[k]
        ctx = self.ctxs[k % len(self.ctxs)]
        self._last_ctx = ctx
        with self._on_stream(k):
            self._render_rank(frame, slot, ctx, k)
            work = self._gather(st, slot, async_gather, k)
            if async_gather:
                return work, st
            if assemble:
                unstripe(st, self.layout, self.frame_outs[k])
                self.frame_out = self.frame_outs[k]
                result = self.frame_out
            else:
                result = st
        if any(s is not None for s in self.streams):
            self.torch.cuda.current_stream().wait_stream(self.streams[k % len(self.streams)])
        return result

    def render_pipelined(self, frame):
        """Throughput form: render frame k into one staging slot and START its all-gather, then assemble the oldest
        frame in flight (whose gather had at least a whole frame time to finish).  RCCL runs on its own stream, ordered
        after the render by an event, so the exchange of frame k overlaps the projection/sort/compositing of frame
        k+1 — on xGMI the gather of 8 stripes costs about as much as a stripe's compute.  Returns the assembled oldest
        frame once the pipeline is full (None before); call flush_all() for the rest."""
        k = self._turn % self.depth
        done = None
        if len(self._pending) >= self.depth:  # slot k is still owned by frame k - depth: retire it first
            done = self._retire()
        self._turn += 1
        st, slot = self.staging[k], self.slot[k]
        ctx = self.ctxs[k % len(self.ctxs)]
        self._last_ctx = ctx
        with self._on_stream(k):
            self._render_rank(frame, slot, ctx, k)
            work = self._gather(st, slot, True, k)
        self._pending.append((work, k))
        if done is None and len(self._pending) >= self.depth:
            done = self._retire()
        return done

    def _retire(self):
        work, k = self._pending.pop(0)
        with self._on_stream(k):
            if work is not None:
                work.wait()  # stream-level wait for the NCCL work; the host does not block on a device backend
            unstripe(self.staging[k], self.layout, self.frame_outs[k])
        self.frame_out = self.frame_outs[k]
        return self.frame_out

    def flush(self):
        """Retire the oldest frame in flight (None if there is none)."""
        return self._retire() if getattr(self, "_pending", None) else None

    def flush_all(self):
        """Retire every frame in flight; returns the newest (None if there was none)."""
        last = None
        while getattr(self, "_pending", None):
            last = self._retire()
        if last is not None and any(s is not None for s in self.streams):
            for s in self.streams:
                self.torch.cuda.current_stream().wait_stream(s)
        return last

    def column_weights(self):
        """Per-tile-column (or row) cost estimate from the last frame: pairs in the stripe's tiles, all-reduced
        so every rank sees the whole frame's profile."""
        torch = self.torch
        n = self._tile_counts()
        w = (n.sum(axis=0) if self.axis == "columns" else n.sum(axis=1)).astype(np.float64)
        c0, c1 = self.layout.cuts[self.rank], self.layout.cuts[self.rank + 1]
        mine = np.zeros_like(w)
        mine[c0:c1] = w[c0:c1]
        t = torch.from_numpy(mine).to(self.device)
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def global_tile_bounds(self):
        """Parity tap: the single-GPU `tile_bounds` rebuilt from the stripes.  Every rank contributes
        the pair count of each of its tiles (T x 4 bytes, one all-reduce of disjoint vectors = an all-gather); the
        ranges are the prefix sums in tile-id order, with quirk Q5/Q6 of gsplat_boundaries.glsl:39-49 on the frame's
        highest populated tile.  Debug path: host arithmetic, one collective."""
        torch = self.torch
        b, d_local = self._local_bounds()
        b = b.astype(np.int64)
        n = np.clip(b[:, 1] - b[:, 0], 0, None)
        tiles = np.arange(self.gx * self.gy)
        coord = (tiles % self.gx) if self.axis == "columns" else (tiles // self.gx)
        c0, c1 = self.layout.cuts[self.rank], self.layout.cuts[self.rank + 1]
        mine = (coord >= c0) & (coord < c1)
        n = np.where(mine, n, 0)
        # this rank's last populated tile may carry the quirk (range never closed, or closed one short): its true
        # length is what is left of the rank's sorted array
        started = np.flatnonzero(mine & ((b[:, 0] > 0) | (b[:, 1] > 0)))
        if started.size:
            t_last = int(started[np.argmax(b[started, 0])])
            n[t_last] = d_local - int(b[t_last, 0])
        t = torch.from_numpy(n).to(self.device)
        self.dist.all_reduce(t, group=self.group)
        n_all = t.cpu().numpy()
        x = np.concatenate([[0], np.cumsum(n_all)[:-1]])
        out = np.zeros((self.gx * self.gy, 2), np.uint32)
        pop = np.flatnonzero(n_all > 0)
        out[pop, 0] = x[pop]
        out[pop, 1] = x[pop] + n_all[pop]
        if pop.size:  # the frame's highest populated tile: never closed, or closed one short when it is tile T-1
            t_star, total = int(pop[-1]), int(n_all.sum())
            out[t_star, 1] = total - 1 if (t_star == self.gx * self.gy - 1 and total > 1) else 0
        return out

    def rebalance(self, per_tile_constant=64.0):
        """New cuts that equalise (pairs + constant per tile) across ranks; identical on every rank."""
        w = self.column_weights()
        other = self.gy if self.axis == "columns" else self.gx
        cuts = balanced_cuts(w + per_tile_constant * other, self.world)
        self.set_cuts(cuts)
        return cuts
