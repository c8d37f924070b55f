#!/usr/bin/env python3
"""Turn the rocpd databases written by tools/profile_gpu.sh into the committed summaries under profiles/.

  python tools/summarize_profile.py gpurun_out/prof_r01_c3 profiles/r01_c3 [config-name]

Writes <prefix>_kernel_stats.md (rocprofv3 --kernel-trace --stats summary: calls, total, average per kernel),
<prefix>_pmc.md (FETCH_SIZE / WRITE_SIZE per kernel, separate passes) and merges the per-launch HBM traffic of
each kernel class into profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of a wide coalesced streaming
read, so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated).  Both counters are in KiB.
"""
import json
import os
import re
import sqlite3
import sys

CLASS_OF = [("project_kernel", "project"), ("frame_plan", "project"), ("tile_sat", "scan"), ("round_filter", "scan"), ("color_kernel", "color"), ("scan_blocks", "scan"), ("emit_sums", "scan"),
            ("emit_kernel", "emit"), ("emit_big", "emit"), ("downsweep_splats", "splat_sort"), ("upsweep_kernel<8>", "splat_sort"),
            ("upsweep", "sort_upsweep"), ("spine", "sort_spine"), ("downsweep_pairs", "sort_downsweep"),
            ("boundaries", "boundaries"), ("tie_long", "boundaries"), ("render_kernel", "render")]


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:60]


def klass(name):
    for pat, k in CLASS_OF:
        if pat in name:
            return k
    return None


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    cfg = sys.argv[3] if len(sys.argv) > 3 else "c3"
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    db = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(prefix + "_kernel_stats.md", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({cfg})\n\n")
        f.write("tree: commit %s (tools/collect_final.sh)\n\n" % os.environ.get("GSPLAT_COMMIT", "?"))
        f.write("command: `rocprofv3 --kernel-trace --stats -- python bench.py --config %s --steps 30 --warmup 5 "
                "--no-cpu-baseline --frames-in-flight 1 --no-host-copy-legs` (tools/profile_gpu.sh): one frame at a time, the regime the "
                "bench line's `roofline` and `ms_per_kernel_class` are measured in; durations in microseconds\n\n" % cfg)
        f.write("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, mn, mx in rows:
            f.write(f"| `{short(name)}` | {calls} | {tot/1e3:.1f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | "
                    f"{100*tot/total:.1f} |\n")
        bj = os.path.join(src, "bench_trace.json")
        if os.path.exists(bj):
            try:
                d = json.loads(open(bj).read().strip().splitlines()[-1])
                f.write("\nbench line of the same run (under the profiler): value = %.1f %s, ms_per_step = %.3f; "
                        "roofline kernel = %s, avg_launch_ms (HIP events) = %.4f\n"
                        % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"]))
            except Exception:
                pass
    dflt = os.path.join(src, "trace_default", "trace_results.db")
    if os.path.exists(dflt):
        d2 = sqlite3.connect(dflt)
        rows2 = d2.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name "
                           "order by sum(duration) desc").fetchall()
        with open(prefix + "_kernel_stats_default_cmd.md", "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats of the DEFAULT bench command ({cfg}, 2 frames in flight)\n\n")
            f.write("tree: commit %s (tools/collect_final.sh)\n\n" % os.environ.get("GSPLAT_COMMIT", "?"))
            f.write("command: `rocprofv3 --kernel-trace --stats -- python bench.py --config %s --no-cpu-baseline --no-host-copy-legs`.  "
                    "Two contexts alternate frames on two streams, so a launch's duration includes the time it shares "
                    "the chip with the other frame's kernels (the 20 timing frames at the end run alone).\n\n" % cfg)
            f.write("| kernel | calls | total us | avg us |\n|---|---|---|---|\n")
            for name, calls, tot, avg in rows2:
                f.write(f"| `{short(name)}` | {calls} | {tot/1e3:.1f} | {avg/1e3:.2f} |\n")
            bj = os.path.join(src, "bench_trace_default.json")
            try:
                d = json.loads(open(bj).read().strip().splitlines()[-1])
                f.write("\nbench line of the same run: value = %.1f %s (sequential_fps = %.1f)\n"
                        % (d["value"], d["unit"], d.get("sequential_fps") or 0.0))
            except Exception:
                pass
    # PMC passes
    rc = summarize_pmc(src, prefix, cfg)
    print("wrote", prefix + "_kernel_stats.md")
    return rc


def frame_of_bench_line(path):
    """{P, D, pairs_round, pair_key_bytes} of the profiled run's own bench line (what the invariants below are checked against)."""
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        ss = d["scene_stats"]
        kb = ss.get("pair_key_bytes")
        return {"P": int(d["config"]["width"]) * int(d["config"]["height"]), "D": int(ss["D"]),
                "pairs_round": [int(x) for x in d.get("pairs_round") or [ss["D"], 0]],
                "sort_passes": int(ss.get("sort_passes", 0)), "pair_key_bytes": kb}
    except Exception:
        return None


def entry_problems(cfg, ent, pixels=None):
    """Why a configuration's entry of pmc_traffic.json cannot be what the kernels moved (empty list: plausible).
    A --pmc pass that did not collect leaves its counter at zero for EVERY kernel (round 5: the WRITE_SIZE pass of c3r and
    c3d, summarised as `2 x FETCH + 0` — a compositor that writes a 33 MB image moving 30 MB); so:
      * the compositor and the pair downsweep both read and write by construction: FETCH_SIZE and WRITE_SIZE > 0;
      * the compositor's heaviest launch writes the image: WRITE_SIZE >= 0.9 x 16 P, traffic >= 16 P;
      * a pair downsweep reads and writes its pairs: traffic >= the bytes of its output, (key + 4) x pairs of the launch
        (where the entry recorded them: `_frame`).
    Used by the summariser (which refuses such a pass) and by tests/test_bench_model.py over the committed file."""
    bad = []
    fr = ent.get("_frame") or {}
    P = pixels if pixels is not None else fr.get("P")
    for k in ("render", "sort_downsweep", "project"):
        e = ent.get(k)
        if not e:
            continue
        if not e.get("fetch_kib", 0.0) > 0.0:
            bad.append(f"{cfg}.{k}: FETCH_SIZE is zero (the pass did not collect)")
        if not e.get("write_kib", 0.0) > 0.0:
            bad.append(f"{cfg}.{k}: WRITE_SIZE is zero (the pass did not collect)")
    r = ent.get("render")
    if r and P:
        if r.get("write_kib", 0.0) * 1024 < 0.9 * 16 * P:
            bad.append(f"{cfg}.render: WRITE_SIZE {r.get('write_kib', 0.0) * 1024 / 1e6:.1f} MB < the {16 * P / 1e6:.1f} MB image it writes")
        if r.get("hbm_bytes_per_launch", 0.0) < 16 * P:
            bad.append(f"{cfg}.render: traffic below the 16 P bytes of the image")
    dsw = ent.get("sort_downsweep")
    if dsw and fr.get("D") and fr.get("sort_passes", 0) > 2:
        pairs = max(fr.get("pairs_round") or [fr["D"]])
        out_bytes = ((fr.get("pair_key_bytes") or 2) + 4) * pairs
        if dsw.get("hbm_bytes_per_launch", 0.0) < out_bytes:
            bad.append(f"{cfg}.sort_downsweep: traffic {dsw.get('hbm_bytes_per_launch', 0.0) / 1e6:.1f} MB below its {out_bytes / 1e6:.1f} MB of output")
    return bad


def summarize_pmc(src, prefix, cfg):
    """FETCH_SIZE / WRITE_SIZE passes -> <prefix>_pmc.md + the configuration's entry of pmc_traffic.json.  A pass that did
    not collect (database missing, counter absent, or zero for every kernel) is a FAILED pass: the summary says so, the
    entry is NOT written (an implausible one that is already there is removed: bench.py then prints "traffic": null), and the
    return value is non-zero so that the collection script fails loudly."""
    pmc, failed = {}, []
    for fn, counter in (("pmc_fetch/fetch_results.db", "FETCH_SIZE"), ("pmc_write/write_results.db", "WRITE_SIZE")):
        path = os.path.join(src, fn)
        if not os.path.exists(path):
            failed.append(f"{counter}: {fn} does not exist")
            continue
        d = sqlite3.connect(path)
        q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name"
        rows = list(d.execute(q, (counter,)))
        if not rows or not any(avg and avg > 0.0 for _, _, avg in rows):
            failed.append(f"{counter}: zero for every kernel of the run ({len(rows)} kernels)")
        for name, calls, avg in rows:
            pmc.setdefault(name, {})[counter] = (calls, avg or 0.0)
    if not pmc and not failed:
        return 0
    traffic = {}
    with open(prefix + "_pmc.md", "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {cfg}\n\n")
        f.write("tree: commit %s (tools/collect_final.sh)\n\n" % os.environ.get("GSPLAT_COMMIT", "?"))
        if failed:
            f.write("**FAILED PASS — not merged into pmc_traffic.json, not a measurement:** " + "; ".join(failed) + "\n\n")
        f.write("Averages per launch, KiB as reported.  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 "
                "every TCC_EA0_RDREQ is a 128-byte line but FETCH_SIZE tallies it at 64 B (MI355X_MICROARCH.md §HBM), "
                "for wide streaming reads AND for gathers of 48-byte / 192-byte records alike, and WRITE_SIZE is exact "
                "for streaming writes and counts 32 B per partially written 64-byte sector — calibrated on known byte "
                "counts, profiles/r02_pmc_calibration.md (round 2; the counters and the correction are unchanged) (tools/pmc_calibrate.hip).\n\n")
        f.write("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB / launch (corrected) |\n|---|---|---|---|\n")
        for name, c in sorted(pmc.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[1])):
            fe = c.get("FETCH_SIZE", (0, 0.0))[1]
            wr = c.get("WRITE_SIZE", (0, 0.0))[1]
            hbm = (2 * fe + wr) * 1024
            f.write(f"| `{short(name)}` | {fe:.1f} | {wr:.1f} | {hbm/1e6:.2f} |\n")
            k = klass(name)
            if k and hbm > traffic.get(k, {}).get("hbm_bytes_per_launch", -1.0):  # the class's heaviest kernel
                traffic[k] = {"hbm_bytes_per_launch": hbm, "fetch_kib": fe, "write_kib": wr, "kernel": short(name)}
    tj = os.path.join(os.path.dirname(prefix) or ".", "pmc_traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    fr = frame_of_bench_line(os.path.join(src, "bench_trace.json"))
    if fr:
        traffic["_frame"] = fr
    problems = failed + entry_problems(cfg, traffic)
    if problems:
        with open(prefix + "_pmc.md", "a") as f:
            f.write("\n**Refused:** " + "; ".join(problems) + "\n")
        old = allt.get(cfg)
        if isinstance(old, dict) and entry_problems(cfg, old, (fr or {}).get("P")):
            del allt[cfg]   # (the entry that is there is no measurement either)
            json.dump(allt, open(tj, "w"), indent=1, sort_keys=True)
        print("PMC PASS REFUSED for", cfg, ":", "; ".join(problems), file=sys.stderr)
        return 3
    allt[cfg] = traffic
    import provenance  # (tools/ is this script's directory)
    allt[cfg]["_csrc_sha256"] = provenance.sha_of_tree()   # the sources these kernels were built from
    allt[cfg]["_collected_at"] = os.environ.get("GSPLAT_COMMIT", "?")
    allt["_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py, summaries under profiles/"
                       + os.path.basename(prefix) + "_pmc.md etc.; commit " + os.environ.get("GSPLAT_COMMIT", "?"))
    json.dump(allt, open(tj, "w"), indent=1, sort_keys=True)
    return 0


    print("wrote", prefix + "_kernel_stats.md")


if __name__ == "__main__":
    sys.exit(main())
