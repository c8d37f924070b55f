"""The N>1 path on CPU: world_size 2 and 3, gloo, 127.0.0.1.  The partition + all-gather + reassembly logic of
godotgaussiansplatting_amd.distributed is the product code under test; the per-rank renderer is a stand-in that
crops the oracle's frame (the HIP renderer's own stripe parity is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, make_case, oracle_frame


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, axis, tmp, exchange=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from godotgaussiansplatting_amd.distributed import StripeRasterizer
        case = make_case(6000, 400, 240, seed=33, scale_n=20000)
        w, h = case["width"], case["height"]
        full = oracle.render_frame(case["records"], oracle_frame(case))
        gx, gy = (w + 15) // 16, (h + 15) // 16

        class Stub(StripeRasterizer):
            """Stand-in renderer: the stripe's pixels of the (sharded-mode) oracle frame."""

            def _apply_stripe(self, b, e):
                self.stripe = (b, e, 0, gy) if self.axis == "columns" else (0, gx, b, e)

            def _render_stripe(self, frame, slot, ctx=None):
                ref = oracle.render_frame(case["records"], oracle_frame(case, stripe=self.stripe))
                self.last = ref
                a, b = self.layout.px_range(self.rank)
                slot.zero_()
                if self.axis == "columns":
                    slot[:, : b - a] = torch.from_numpy(ref["image"][:, a:b])
                else:
                    slot[: b - a] = torch.from_numpy(ref["image"][a:b])

            def _tile_counts(self):
                bb = self.last["bounds"].astype(np.int64)
                return np.clip(bb[:, 1] - bb[:, 0], 0, None).reshape(gy, gx)

            def _local_bounds(self):
                return self.last["bounds"], int(self.last["D"])

            # exchange_last_tile path (contexts that skip whole blocks of the scene): each rank reports a
            # stripe-local "highest populated tile + 1", the 4-byte all-reduce(MAX) must hand every rank the frame's
            def _render_begin(self, frame, ctx, word):
                self.began = getattr(self, "began", 0) + 1
                word.fill_(1000 + 7 * self.rank)

            def _render_end(self, slot, ctx, word):
                a, b = self.layout.px_range(self.rank)
                nonempty = [r for r in range(self.world) if self.layout.px_range(r)[1] > self.layout.px_range(r)[0]]
                assert int(word.item()) == 1000 + 7 * max(nonempty), (int(word.item()), nonempty)
                self._render_stripe(None, slot, ctx)

        sr = Stub(None, w, h, rank, world, axis=axis, device=torch.device("cpu"), exchange_last_tile=exchange)
        out = sr.render(None).numpy().copy()
        np.testing.assert_array_equal(out, full["image"])      # every rank holds the whole frame, bit-exact
        assert sr.staging[0].shape[-1] == 3                    # 12 bytes per pixel on the wire (alpha == 1.0)
        # the single-GPU tile_bounds tap rebuilt from the stripes' per-tile counts (T x 4 bytes all-gathered)
        np.testing.assert_array_equal(sr.global_tile_bounds(), full["bounds"])
        cuts0 = list(sr.layout.cuts)
        cuts1 = sr.rebalance(per_tile_constant=8.0)
        gathered = [None] * world
        dist.all_gather_object(gathered, cuts1)
        assert all(c == cuts1 for c in gathered)               # same partition everywhere
        assert cuts1[0] == 0 and cuts1[-1] == (gx if axis == "columns" else gy)
        out = sr.render(None).numpy().copy()
        np.testing.assert_array_equal(out, full["image"])
        # pipelined form used by bench.py: frame k's gather overlaps frame k+1's render
        assert sr.render_pipelined(None) is None        # pipeline filling (depth 2)
        for _ in range(4):
            prev = sr.render_pipelined(None)
            np.testing.assert_array_equal(prev.numpy(), full["image"])
        np.testing.assert_array_equal(sr.flush_all().numpy(), full["image"])
        assert sr.flush() is None and sr.flush_all() is None
        # async gather + explicit assembly
        work, st = sr.render(None, async_gather=True)
        work.wait()
        sr._turn = 0
        from godotgaussiansplatting_amd.distributed import unstripe
        canvas = torch.zeros(h, w, 4)
        canvas[..., 3] = 1.0
        np.testing.assert_array_equal(unstripe(st, sr.layout, canvas).numpy(), full["image"])
        if exchange:
            assert sr.began >= 8  # every frame above went through begin / all-reduce / end
            # a rank without tiles still takes part in the collective: 1-tile-wide stripes for rank 0 only
            n_units = gx if axis == "columns" else gy
            sr.set_cuts([0] + [n_units] * world)
            np.testing.assert_array_equal(sr.render(None).numpy(), full["image"])
        with open(os.path.join(tmp, f"ok_{rank}"), "w") as f:
            f.write(f"{cuts0} -> {cuts1}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,axis", [(2, "columns"), (3, "rows")])
def test_stripe_gather_gloo(world, axis, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, axis, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok_{r}" for r in range(world)]


def test_stripe_gather_with_last_tile_exchange_gloo(tmp_path):
    """The begin / all-reduce(MAX) / end form used with GSPLAT_FLAG_BLOCK_CULL contexts, world size 3."""
    port = _free_port()
    mp.spawn(_worker, args=(3, port, "columns", str(tmp_path), True), nprocs=3, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok_{r}" for r in range(3)]
