#!/usr/bin/env python3
"""Summarise tools/pmc_calibrate.sh: counters per launch next to the bytes each micro-kernel asks for.

  python tools/pmc_calibrate_report.py gpurun_out/pmc_cal profiles/r02_pmc_calibration.md
"""
import json
import os
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    exp = json.loads(open(os.path.join(src, "expected.json")).read().strip().splitlines()[-1])
    vals = {}
    for i in range(1, 5):
        for root, _, files in os.walk(os.path.join(src, f"pass{i}")):
            for fn in files:
                if fn.endswith("_results.db"):
                    db = sqlite3.connect(os.path.join(root, fn))
                    q = "select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"
                    for name, counter, avg in db.execute(q):
                        key = name.split("(")[0].replace("void ", "")
                        vals.setdefault(key, {})[counter] = avg
    with open(out, "w") as f:
        f.write("# rocprofv3 PMC calibration on known byte counts (tools/pmc_calibrate.hip, 3 GiB footprint)\n\n")
        f.write("Averages per launch.  FETCH_SIZE / WRITE_SIZE in KiB as reported; RDREQ / 32B / BUBBLE / WRREQ / 64B are the "
                "raw TCC_EA0 request counters they derive from.  `asked` = bytes the kernel's loads or stores name; "
                "`line-granular` = the same at 64-byte sector granularity for the gathers (every record start is random).\n\n")
        f.write("| kernel | asked MB | FETCH_SIZE MB | x | WRITE_SIZE MB | x | RDREQ | RDREQ_32B | BUBBLE | WRREQ | WRREQ_64B |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, e in exp.items():
            v = next((vv for kk, vv in vals.items() if kk.startswith(k.split("<")[0]) and (("<" not in k) or k.split("<")[1].rstrip(">") in kk)), {})
            asked = e.get("read", e.get("write", 0))
            fe = v.get("FETCH_SIZE", 0.0) * 1024
            wr = v.get("WRITE_SIZE", 0.0) * 1024
            f.write(f"| `{k}` | {asked/1e6:.1f} | {fe/1e6:.1f} | {fe/asked if 'read' in e else 0:.3f} | {wr/1e6:.1f} | "
                    f"{wr/asked if 'write' in e else 0:.3f} | {v.get('TCC_EA0_RDREQ_sum', 0):.0f} | {v.get('TCC_EA0_RDREQ_32B_sum', 0):.0f} | "
                    f"{v.get('TCC_BUBBLE_sum', 0):.0f} | {v.get('TCC_EA0_WRREQ_sum', 0):.0f} | {v.get('TCC_EA0_WRREQ_64B_sum', 0):.0f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
