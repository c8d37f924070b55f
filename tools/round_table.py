#!/usr/bin/env python3
"""The per-configuration table of DESIGN.md §7 / README.md from profiles/<round>_bench_*.json (so the documents quote the
files, not a transcription).  usage: python tools/round_table.py r05"""
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {"c1": "c1 100 k, 720p, deg 0", "c2": "c2 1 M, 1080p, deg 0", "c3": "**c3 6.13 M, 1080p, deg 3**", "c3m": "c3m = c3 at D/N 4.1",
         "c3d": "c3d = c3 at D/N 9.4", "c3r": "**c3r = capture-shaped** (D/N 7.9)", "c4": "c4 5.83 M, 4K, deg 3", "c5": "c5 30 M, 4K, deg 0"}
print("| config | value (2 in flight) | orbit | one at a time | fps_with_d2h (sync) | frame GPU ms (proj+emit / sort / bounds / render) | pairs A + B of D | roofline.frac (kernel) | parity |")
print("|---|---|---|---|---|---|---|---|---|")
for c, label in names.items():
    p = os.path.join(ROOT, "profiles", f"{R}_bench_{c}.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    pp = d["ms_per_pass"]
    pr = d["pairs_round"]
    r = d.get("roofline", {})
    pc = d.get("parity_check") or {}
    ok = all(pc.get(k) for k in ("rgba_bit_exact", "bounds_equal", "keys_equal", "values_equal")) if pc else None
    print(f"| {label} | {d['value']:.0f} | {d.get('value_moving_camera', 0):.0f} | {d['sequential_fps']:.0f} | "
          f"{d.get('fps_with_d2h', 0):.0f} ({d.get('fps_with_sync_d2h', 0):.0f}) | {d['frame_ms_gpu']['p50']:.3f} ({pp['projection']:.3f} / "
          f"{pp['sort']:.3f} / {pp['boundaries']:.3f} / {pp['render']:.3f}) | {pr[0] / 1e6:.2f} M + {pr[1] / 1e6:.2f} M of {d['scene_stats']['D'] / 1e6:.1f} M | "
          f"{r.get('frac', 0):.3f} ({r.get('kernel')}; traffic {'stale' if r.get('traffic_stale') else (round(r['traffic'] / 1e6) if r.get('traffic') else None)} MB) | "
          f"{'array_equal' if ok else ok} |")
