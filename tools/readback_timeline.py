#!/usr/bin/env python3
"""Does the copy of frame k overlap the kernels of frame k + 1?  Reads a rocprofv3 --kernel-trace --memory-copy-trace
database of tools/d2h_probe.py and prints, for the last frames of the pipelined loop, when the copy (readback_kernel or
the runtime's copy) and the next frame's kernels ran.  usage: python tools/readback_timeline.py <dir>"""
import glob
import os
import re
import sqlite3
import sys

paths = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)
db = sqlite3.connect(paths[0])


def cols(t):
    return [r[1] for r in db.execute(f"PRAGMA table_info({t})")]


kc = cols("kernels")
s_col = "start" if "start" in kc else [c for c in kc if "start" in c][0]
e_col = "end" if "end" in kc else [c for c in kc if "end" in c][0]
extra = [c for c in ("queue_id", "stream_id") if c in kc]
rows = db.execute(f"select name, {s_col}, {e_col}" + "".join(", " + c for c in extra) + f" from kernels order by {s_col}").fetchall()


def short(name):
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


ev = [(short(r[0]), r[1], r[2]) + tuple(r[3:]) for r in rows]
try:
    mc = cols("memory_copies")
    ms = "start" if "start" in mc else [c for c in mc if "start" in c][0]
    me = "end" if "end" in mc else [c for c in mc if "end" in c][0]
    for r in db.execute(f"select {ms}, {me} from memory_copies"):
        if r[1] - r[0] > 200_000:   # the frame copies (33 MB), not the small ones
            ev.append(("MEMORY_COPY", r[0], r[1]))
except Exception as e:  # noqa: BLE001
    print("no memory_copies:", e)
ev.sort(key=lambda x: x[1])
copies = [e for e in ev if e[0] in ("readback_kernel", "MEMORY_COPY") or "copyBuffer" in e[0] and e[2] - e[1] > 200_000]
print(len(copies), "frame copies;", "columns", extra)
for c in copies[-6:-2]:
    inside = [e for e in ev if e is not c and e[1] < c[2] and e[2] > c[1]]
    names = {}
    for e in inside:
        ov = min(e[2], c[2]) - max(e[1], c[1])
        names[e[0]] = names.get(e[0], 0) + ov
    print(f"copy {c[0]} {(c[2]-c[1])/1e3:.0f} us" + (f" queue {c[3:]}" if len(c) > 3 else "") + ": overlapping kernels (us) " +
          ", ".join(f"{k} {v/1e3:.0f}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]))
    nxt = [e for e in ev if e[0] == "project_kernel" and e[1] > c[1]][:1]
    if nxt:
        print(f"   next project_kernel starts {(nxt[0][1]-c[1])/1e3:.0f} us after the copy's start, lasts {(nxt[0][2]-nxt[0][1])/1e3:.0f} us" + (f" queue {nxt[0][3:]}" if len(nxt[0]) > 3 else ""))
