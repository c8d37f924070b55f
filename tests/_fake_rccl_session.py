"""Run by tests/test_gpu_parity.py::test_group_exchange_with_several_members_on_a_stand_in_rccl in a process of its own,
with GSPLAT_RCCL_LIB pointing at tests/native/fake_rccl.hip's library (a test double: every rank in this process, on
this one GPU).  gsplat_group_render with MORE THAN ONE member — stripe geometry, the 4-byte all-reduce, the
all-gather-v in both forms, unequal cuts, frames in flight on several groups — against the single-GPU oracle frame."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import hip_frame, make_case, oracle_frame  # noqa: E402
import oracle  # noqa: E402
from godotgaussiansplatting_amd import _lib, capi  # noqa: E402

assert os.environ.get("GSPLAT_RCCL_LIB", "").endswith(".so")
assert "test_hooks" in os.environ.get("GSPLAT_LIB", ""), "needs the test build of the library (build.build_test_hooks)"

n, w, h = 30000, 1000, 540
case = make_case(n, w, h, seed=651, sh_degree=2, scale_n=3000)
REF = oracle.render_frame(case["records"], oracle_frame(case), capacity=40 * n)["image"]
frame = hip_frame(case)
gx, gy = (w + 15) // 16, (h + 15) // 16
checked = 0


TIES_REF = {}


def members(world, cull, ties=False):
    """ties: the flags bench.py --gpus N really runs with — GSPLAT_FLAG_BLOCK_CULL | GSPLAT_FLAG_TIES_STORAGE_ORDER on a
    re-laid-out scene (16-bit pair keys, the one-pass pair sort on a stripe's <= 1024 local tiles, no tie repair)."""
    flags = (capi.FLAG_BLOCK_CULL if cull else 0) | (capi.FLAG_TIES_STORAGE_ORDER if ties else 0)
    kw = dict(key_budget_factor=40, flags=flags)
    owner = capi.Context(n, w, h, **kw)
    owner.upload_splats(case["records"])
    if cull or ties:
        owner.finalize_scene()
    if ties and "image" not in TIES_REF:
        # the storage-order contract's frame = the default contract's frame of the scene UPLOADED in storage order
        ids = owner.read_slot_ids()
        TIES_REF["image"] = oracle.render_frame(case["records"][ids], oracle_frame(case), capacity=40 * n)["image"]
    return [owner] + [owner.view(**kw) for _ in range(world - 1)]


def same(ctxs, what, want=None):
    global checked
    ref = want if want is not None else REF
    for r, c in enumerate(ctxs):
        c.synchronize()
        got = c.read_image()
        if not np.array_equal(got, ref):
            bad = np.argwhere((got != ref).any(axis=2))
            raise SystemExit(f"{what}: member {r} differs from the single-GPU frame at {len(bad)} pixels, first {bad[0]}")
        checked += 1


# 1. one thread drives every member (gsplat_group_create_local): 2 and 3 members, both axes, both gather forms,
#    with and without block culling (= with and without the all-reduce), equal and unequal stripes, an empty stripe
#    — with 12 bytes per pixel on the wire (default: RGB through the staging buffer, alpha rebuilt) and with 16 (rgba: row
#    stripes in place)
for gather, pixels in (("p2p", "rgb"), ("broadcast", "rgb"), ("p2p", "rgba")):
    os.environ["GSPLAT_GROUP_GATHER"] = gather
    os.environ["GSPLAT_GROUP_PIXELS"] = pixels
    for world in ((2, 3, 8) if (gather, pixels) == ("p2p", "rgb") else (2, 3)):
        for axis, extent in ((capi.STRIPE_ROWS, gy), (capi.STRIPE_COLUMNS, gx)):
            for cull in (False, True):
                ctxs = members(world, cull)
                what = f"local {gather} {pixels} world={world} axis={axis} cull={cull}"
                rng = np.random.default_rng(1000 * world + 10 * axis + cull)
                with capi.Group.local(ctxs, axis=axis) as g:
                    assert g.exchanges_last_tile() == cull, what
                    for _ in range(2):
                        g.render(frame)
                    same(ctxs, what)
                    inner = sorted(int(x) for x in rng.choice(np.arange(1, extent), world - 1, replace=False))
                    g.set_cuts([0] + inner + [extent])     # unequal stripes: all-gather-v
                    g.render(frame)
                    same(ctxs, what + f" cuts={inner}")
                    inner[-1] = inner[0] if world > 2 else extent   # a member (world=2: the last) without tiles
                    g.set_cuts([0] + sorted(inner) + [extent])
                    g.render(frame)
                    same(ctxs, what + f" empty stripe {sorted(inner)}")
                for c in reversed(ctxs):
                    c.close()
os.environ["GSPLAT_GROUP_GATHER"] = "p2p"
os.environ["GSPLAT_GROUP_PIXELS"] = "rgb"

# 2. one thread per rank (gsplat_group_create: a blocking rendezvous on a unique id), two groups per rank = two frames in
#    flight on communicators of their own, as bench.py --gpus N runs them
#    — `frames` = contexts per rank = groups in flight (bench.py --gpus N: four); cull: per RANK, so ranks may DISAGREE
#    (round 4 decided the all-reduce per frame from each rank's own state: a rank whose scene was finalized while its
#    peer's was not entered ncclAllReduce alone — a hang; now the ranks agree once, inside gsplat_group_create)
def rank_threads(world, frames, cull_of_rank, what, ties=False, ties_of_rank=None):
    ties_of_rank = ties_of_rank or [ties] * world
    # rank r: `frames` contexts on its own copy of the scene
    rings = [members(frames, cull_of_rank[r], ties_of_rank[r]) for r in range(world)]
    mixed = len(set(ties_of_rank)) > 1
    refused = [False] * world
    ids = [capi.group_unique_id() for _ in range(frames)]
    errors, agreed = [], [None] * world
    barrier = threading.Barrier(world)

    def rank_main(r):
        try:
            if mixed:
                # ranks that differ in GSPLAT_FLAG_TIES_STORAGE_ORDER: gsplat_group_create fails on EVERY rank (the flag
                # rides in the creation-time all-reduces), nobody is left inside a collective
                try:
                    capi.Group(rings[r][0], ids[0], r, world, capi.STRIPE_ROWS)
                except _lib.GsplatError as e:
                    refused[r] = "TIES_STORAGE_ORDER" in str(e)
                return
            groups = [capi.Group(rings[r][k], ids[k], r, world, capi.STRIPE_ROWS) for k in range(frames)]
            agreed[r] = [g.exchanges_last_tile() for g in groups]
            for f in range(3 * frames):
                groups[f % frames].render(frame)
            for c in rings[r]:
                c.synchronize()
            inner = sorted({max(1, (gy * (k + 1)) // (world + 1)) for k in range(world - 1)})
            while len(inner) < world - 1:                       # (a degenerate cut: a member without tiles)
                inner.append(inner[-1])
            for g in groups:                                    # unequal stripes, the same on every rank
                g.set_cuts([0] + inner + [gy])
            barrier.wait()
            for f in range(2 * frames):
                groups[f % frames].render(frame)
            for c in rings[r]:
                c.synchronize()
            for g in groups:
                g.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    if errors or any(t.is_alive() for t in threads):
        raise SystemExit(f"rank threads ({what}): {errors or 'still running'}")
    if mixed:
        if not all(refused):
            raise SystemExit(f"{what}: gsplat_group_create must refuse mixed tie flags on every rank: {refused}")
        for r in range(world):
            for c in reversed(rings[r]):
                c.close()
        return
    want = any(cull_of_rank)
    if any(a != [want] * frames for a in agreed):
        raise SystemExit(f"{what}: the ranks did not agree on the last-tile exchange: {agreed}, expected {want}")
    for r in range(world):
        same(rings[r], f"rank form {what} rank={r}", TIES_REF["image"] if ties else None)
        for c in reversed(rings[r]):
            c.close()


rank_threads(2, 2, [False, False], "2 ranks x 2 frames in flight")
rank_threads(2, 2, [True, True], "2 ranks x 2 in flight, block culling")
rank_threads(2, 2, [True, False], "2 ranks that DISAGREE: rank 0 finalized + culling, rank 1 not")
rank_threads(3, 2, [False, True, False], "3 ranks, only the middle one culls")
rank_threads(4, 4, [True] * 4, "bench topology: 4 ranks x 4 groups in flight, Morton + culling")
rank_threads(8, 4, [True] * 8, "bench topology: 8 ranks x 4 groups in flight, Morton + culling")
# ... and with the flags bench.py --gpus N really passes (MULTI_FLAGS = BLOCK_CULL | TIES_STORAGE_ORDER: 16-bit keys, one-pass
# pair sort, no tie repair), three groups in flight as the bench keeps them — against the ORACLE's frame of the scene in
# storage order
rank_threads(2, 3, [True] * 2, "bench flags: 2 ranks x 3 in flight, cull + ties in storage order", ties=True)
rank_threads(8, 3, [True] * 8, "bench flags: 8 ranks x 3 in flight, cull + ties in storage order", ties=True)
rank_threads(2, 1, [True] * 2, "ranks that differ in the tie flag are refused", ties_of_rank=[True, False])
# the one-thread local form with the same flags, and its refusal of mixed members
ctxs = members(3, True, ties=True)
with capi.Group.local(ctxs, axis=capi.STRIPE_ROWS) as g:
    for _ in range(2):
        g.render(frame)
    same(ctxs, "local cull + ties world=3", TIES_REF["image"])
odd = ctxs[0].view(key_budget_factor=40, flags=capi.FLAG_BLOCK_CULL)
try:
    capi.Group.local([ctxs[0], odd], axis=capi.STRIPE_ROWS).close()
    raise SystemExit("a local group of members with different tie flags must be refused")
except _lib.GsplatError as e:
    assert "TIES_STORAGE_ORDER" in str(e), str(e)
odd.close()
for c in reversed(ctxs):
    c.close()

# 3. batched frames through the group: gsplat_group_render_batch — B consecutive frames (B cameras) of every member through one
#    launch sequence, ONE all-reduce of B words, ONE all-gather-v in which a member's B stripes are one message per peer.
#    Members are batch contexts (views of a scene owner that is not a member itself); bench flags (cull + ties in storage order)
from conftest import godot_perspective  # noqa: E402
from godotgaussiansplatting_amd import scenes  # noqa: E402


def camera_frames(count):
    out_frames, out_cases = [], []
    for k in range(count):
        ang = 0.3 + 0.25 * k
        cam = scenes.look_at_camera((5.0 * np.sin(ang), 0.4, 5.0 * np.cos(ang)))
        vp = oracle.pack_camera(cam.xform12(), godot_perspective(cam.fov, w / h, cam.near, cam.far))
        pos = np.array([-cam.origin[0], -cam.origin[1], cam.origin[2]], np.float32)
        ck = dict(case, vp=vp, cam_pos=pos, camera=cam)
        out_cases.append(ck)
        out_frames.append(hip_frame(ck))
    return out_frames, out_cases


def batch_members(world, batch):
    flags = capi.FLAG_BLOCK_CULL | capi.FLAG_TIES_STORAGE_ORDER
    owner = capi.Context(n, w, h, key_budget_factor=40, flags=flags)
    owner.upload_splats(case["records"])
    owner.finalize_scene()
    return owner, [owner.view(key_budget_factor=40, flags=flags, batch=batch) for _ in range(world)]


bframes, bcases = camera_frames(3)
owner, ctxs = batch_members(3, 4)
ids = owner.read_slot_ids()
brefs = [oracle.render_frame(case["records"][ids], oracle_frame(ck), capacity=40 * n)["image"] for ck in bcases]


def same_batch(members_, count, what):
    global checked
    for r, c in enumerate(members_):
        c.synchronize()
        imgs = c.read_batch_images(count)
        for k in range(count):
            if not np.array_equal(imgs[k], brefs[k]):
                bad = np.argwhere((imgs[k] != brefs[k]).any(axis=2))
                raise SystemExit(f"{what}: member {r} frame {k} differs from the oracle frame at {len(bad)} pixels, first {bad[0]}")
            checked += 1


for gather, pixels in (("p2p", "rgb"), ("broadcast", "rgb"), ("p2p", "rgba")):
    os.environ["GSPLAT_GROUP_GATHER"] = gather
    os.environ["GSPLAT_GROUP_PIXELS"] = pixels
    with capi.Group.local(ctxs, axis=capi.STRIPE_ROWS) as g:
        assert g.exchanges_last_tile()
        for _ in range(2):
            g.render_batch(bframes)
        same_batch(ctxs, 3, f"local batch of 3, {gather} {pixels}")
        g.set_cuts([0, 9, 10, gy])
        g.render_batch(bframes[:2])
        same_batch(ctxs, 2, f"local batch of 2, unequal cuts, {gather} {pixels}")
        g.render(bframes[1])                                    # a plain frame through the same group and members
        for c in ctxs:
            c.synchronize()
            if not np.array_equal(c.read_image(), brefs[1]):
                raise SystemExit(f"plain frame after batches ({gather} {pixels}) differs")
            checked += 1
os.environ["GSPLAT_GROUP_GATHER"] = "p2p"
os.environ["GSPLAT_GROUP_PIXELS"] = "rgb"
for c in reversed(ctxs):
    c.close()
owner.close()

# ... and one thread per rank, two groups (= two batches) in flight per rank, 4 ranks, columns this time
def batch_rank_threads(world, in_flight, batch):
    rings = [batch_members(in_flight, batch) for _ in range(world)]
    gids = [capi.group_unique_id() for _ in range(in_flight)]
    errors = []

    def rank_main(r):
        try:
            groups = [capi.Group(rings[r][1][k], gids[k], r, world, capi.STRIPE_COLUMNS) for k in range(in_flight)]
            for f in range(3 * in_flight):
                groups[f % in_flight].render_batch(bframes[:batch])
            for c in rings[r][1]:
                c.synchronize()
            # (compared before the groups go: gsplat_group_destroy hands the contexts back as plain, unstriped ones)
            same_batch(rings[r][1], batch, f"rank form, {world} ranks x {in_flight} batches of {batch} in flight, rank {r}")
            barrier.wait()
            for g in groups:
                g.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    barrier = threading.Barrier(world)
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    if errors or any(t.is_alive() for t in threads):
        raise SystemExit(f"batch rank threads: {errors or 'still running'}")
    for r in range(world):
        for c in reversed(rings[r][1]):
            c.close()
        rings[r][0].close()


batch_rank_threads(4, 2, 3)

print(f"FAKE_RCCL_SESSION_OK {checked} frames compared")
