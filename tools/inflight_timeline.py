#!/usr/bin/env python3
"""How do the kernels of several frames in flight share the chip?  Reads a rocprofv3 --kernel-trace database of
tools/host_submit_probe.py (a stripe rank rendered with R frames in flight) and prints, for a window in the steady state of
each R: how long every kernel class takes compared with running alone (R = 1), how much of the wall time at least one kernel
is running, and how many run at once on average.  usage: python tools/inflight_timeline.py <dir>"""
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0])
kc = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
s_col = "start" if "start" in kc else [c for c in kc if "start" in c][0]
e_col = "end" if "end" in kc else [c for c in kc if "end" in c][0]
q_col = "queue_id" if "queue_id" in kc else None
rows = db.execute(f"select name, {s_col}, {e_col}" + (f", {q_col}" if q_col else "") + f" from kernels order by {s_col}").fetchall()


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:30]


ev = [(short(r[0]), r[1], r[2], r[3] if q_col else 0) for r in rows]
# the probe renders R = 1, 2, 3, 4, 6 in turn, 120 timed frames each after warm-up: split the trace at render_kernel counts
renders = [i for i, e in enumerate(ev) if e[0] == "render_kernel"]
print("kernels", len(ev), "compositor launches", len(renders), "queues", sorted({e[3] for e in ev}))
# windows: find phases by the number of distinct queues active in a sliding window of 40 compositor launches
phase_edges = []
cur = None
for k in range(0, len(renders) - 40, 20):
    qs = len({ev[i][3] for i in range(renders[k], renders[k + 40])})
    if qs != cur:
        phase_edges.append((k, qs))
        cur = qs
print("phases (first compositor launch, queues in flight):", phase_edges)
base = {}
for idx, (k0, qs) in enumerate(phase_edges):
    k1 = phase_edges[idx + 1][0] if idx + 1 < len(phase_edges) else len(renders) - 1
    if k1 - k0 < 60:
        continue
    a, b = renders[k0 + (k1 - k0) // 2], renders[min(k1 - 5, k0 + (k1 - k0) // 2 + 45)]   # ~45 frames in the phase's second half
    win = ev[a:b]
    t0, t1 = win[0][1], max(e[2] for e in win)
    per = defaultdict(list)
    for e in win:
        per[e[0]].append(e[2] - e[1])
    pts = sorted([(e[1], 1) for e in win] + [(e[2], -1) for e in win])
    busy, conc_time, depth, last = 0, 0, 0, t0
    for t, d in pts:
        if depth > 0:
            busy += t - last
            conc_time += (t - last) * depth
        depth += d
        last = t
    frames = sum(1 for e in win if e[0] == "render_kernel")
    print(f"\n== {qs} queue(s) in flight: {frames} frames in {(t1 - t0) / 1e3:.0f} us = {(t1 - t0) / 1e3 / frames:.1f} us per frame; some kernel running "
          f"{100.0 * busy / (t1 - t0):.0f} % of the time, {conc_time / max(busy, 1):.2f} kernels at once while any runs")
    line = []
    for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(ds) / len(ds) / 1e3
        if qs == 1:
            base[name] = avg
        line.append(f"{name} {avg:.1f}" + (f" (x{avg / base[name]:.2f})" if name in base and qs != 1 else ""))
    print("   avg us per launch: " + ", ".join(line[:14]))
    print(f"   sum of kernel time per frame: {sum(sum(ds) for ds in per.values()) / 1e3 / frames:.1f} us")
