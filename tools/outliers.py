#!/usr/bin/env python3
"""Which launches of a kernel are slow, and what shared the chip with them?  Reads a rocprofv3 --kernel-trace
(--memory-copy-trace) database written on the GPU box and prints a compact report (the database itself stays there).

    python tools/outliers.py <dir with *_results.db> [kernel substring = project_kernel] [slow factor = 1.5]
"""
import glob
import os
import sqlite3
import sys


def cols(db, table):
    return [r[1] for r in db.execute(f"PRAGMA table_info({table})")]


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "project_kernel"
    factor = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
    paths = glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True)
    if not paths:
        print("no database under", src)
        return 1
    db = sqlite3.connect(paths[0])
    kc = cols(db, "kernels")
    print("kernels columns:", kc)
    s_col = "start" if "start" in kc else [c for c in kc if "start" in c][0]
    e_col = "end" if "end" in kc else [c for c in kc if "end" in c][0]
    extra = [c for c in ("stream_id", "queue_id", "stream", "queue") if c in kc]
    rows = db.execute(f"select name, {s_col}, {e_col}" + "".join(", " + c for c in extra) + " from kernels order by " + s_col).fetchall()
    import re

    def short(name):
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd_rocclr_\w+)", name)
        return m.group(1) if m else name[:50]
    ks = [(short(r[0]), r[1], r[2]) + tuple(r[3:]) for r in rows]
    copies = []
    try:
        mc = cols(db, "memory_copies")
        ms = "start" if "start" in mc else [c for c in mc if "start" in c][0]
        me = "end" if "end" in mc else [c for c in mc if "end" in c][0]
        nm = "name" if "name" in mc else mc[0]
        sz = [c for c in mc if "size" in c or "bytes" in c]
        copies = db.execute(f"select {nm}, {ms}, {me}" + (", " + sz[0] if sz else "") + " from memory_copies").fetchall()
        print("memory_copies:", len(copies), "columns", mc)
    except Exception as e:  # noqa: BLE001
        print("no memory_copies table:", e)
    mine = [k for k in ks if pat in k[0]]
    if not mine:
        print("no kernel matches", pat)
        return 1
    durs = sorted(k[2] - k[1] for k in mine)
    lo = durs[0]
    print(f"{pat}: {len(mine)} launches, min {lo/1e3:.1f} us, median {durs[len(durs)//2]/1e3:.1f}, max {durs[-1]/1e3:.1f}")
    slow = [k for k in mine if (k[2] - k[1]) > factor * lo]
    print(f"{len(slow)} launches slower than {factor} x min; what overlapped them (summed overlap, us):")
    agg, per_launch = {}, []
    for k in slow:
        s, e = k[1], k[2]
        here = {}
        for o in ks:
            if o[2] <= s or o[1] >= e or o is k:
                continue
            ov = min(e, o[2]) - max(s, o[1])
            here[o[0]] = here.get(o[0], 0) + ov
        for c in copies:
            if c[2] <= s or c[1] >= e:
                continue
            ov = min(e, c[2]) - max(s, c[1])
            nm = "memcpy:" + str(c[0]) + (f"[{c[3]/1e6:.0f}MB]" if len(c) > 3 and c[3] else "")
            here[nm] = here.get(nm, 0) + ov
        for n, v in here.items():
            agg[n] = agg.get(n, 0) + v
        per_launch.append(((e - s) / 1e3, sorted(here.items(), key=lambda kv: -kv[1])[:3]))
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:10]:
        print(f"  {n}: {v/1e3:.0f}")
    alone = sum(1 for d, top in per_launch if not top)
    print(f"slow launches with NOTHING overlapping: {alone}")
    for d, top in sorted(per_launch, key=lambda x: -x[0])[:12]:
        print(f"  {d:.0f} us: " + ", ".join(f"{n} {v/1e3:.0f}" for n, v in top))
    # position of the slow launches in the run (which bench leg): index among all launches of the kernel
    idx = [i for i, k in enumerate(mine) if (k[2] - k[1]) > factor * lo]
    print("indices of slow launches (of", len(mine), "):", idx[:60], "..." if len(idx) > 60 else "")
    return 0


if __name__ == "__main__":
    sys.exit(main())
