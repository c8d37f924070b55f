"""Test-side restatement of the compositor's tile schedule (csrc/gsplat_internal.h: order_layout, projection.hip:
schedule_tiles) — which tile workgroup b of render_kernel takes.  Changes the schedule only, never the image."""
import numpy as np

ORDER_CLASSES = 32
EMPTY = 0xFFFFFFFF


def order_class(staged):
    """0 = heaviest: half a staging batch (128 pairs) per class, capped."""
    c = (np.asarray(staged, np.int64) + 127) >> 7
    return ORDER_CLASSES - 1 - np.minimum(c, ORDER_CLASSES - 1)


def order_layout(sw, sh):
    bw = min(8, max(sw, 1))
    bh = 1 if sh < 2 else 2
    nbx = (sw + bw - 1) // bw
    if nbx % 8 == 0:
        nbx += 1          # one virtual, empty block column: block columns must not line up with the XCDs
    nby = (sh + bh - 1) // bh
    nblocks = nbx * nby
    per_xcd = ((nblocks + 7) // 8) * bw * bh
    return dict(bw=bw, bh=bh, nbx=nbx, nby=nby, nblocks=nblocks, per_xcd=per_xcd, entries=8 * per_xcd)


def xcd_lists(rect, gx):
    """rect = (sx0, sx1, sy0, sy1) in tiles.  Per XCD: the tile ids (EMPTY for the slots of partial / virtual blocks) in
    enumeration order — block after block (block B of the row-major block grid belongs to XCD B % 8), column-major
    inside a block."""
    sx0, sx1, sy0, sy1 = rect
    lay = order_layout(sx1 - sx0, sy1 - sy0)
    bsz = lay["bw"] * lay["bh"]
    lists = []
    for x in range(8):
        tiles = np.full(lay["per_xcd"], EMPTY, np.int64)
        for j in range(lay["per_xcd"]):
            q, sl = divmod(j, bsz)
            B = x + 8 * q
            by_, bx_ = divmod(B, lay["nbx"])
            tx, ty = sx0 + bx_ * lay["bw"] + sl // lay["bh"], sy0 + by_ * lay["bh"] + sl % lay["bh"]
            if B < lay["nblocks"] and tx < sx1 and ty < sy1:
                tiles[j] = ty * gx + tx
        lists.append(tiles)
    return lay, lists


def expected_order(prev_staged, rect, gx, mode="xcd"):
    """The table scan_blocks_kernel builds from the previous frame's staged counts."""
    sx0, sx1, sy0, sy1 = rect
    if mode == "lpt":
        tiles = np.array([y * gx + x for y in range(sy0, sy1) for x in range(sx0, sx1)], np.int64)
        return tiles[np.argsort(order_class(prev_staged[tiles]), kind="stable")].astype(np.uint32)
    lay, lists = xcd_lists(rect, gx)
    table = np.full(lay["entries"], EMPTY, np.int64)
    for x, tiles in enumerate(lists):
        cls = np.where(tiles == EMPTY, ORDER_CLASSES, order_class(prev_staged[np.where(tiles == EMPTY, 0, tiles)]))
        table[x::8] = tiles[np.argsort(cls, kind="stable")]
    return table.astype(np.uint32)
