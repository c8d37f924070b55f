#!/usr/bin/env python3
"""Per-kernel averages of every counter found in the rocpd databases under <dir>/pass*/ (tools/pmc_gpu.sh)."""
import glob, os, re, sqlite3, sys
from collections import defaultdict
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
data = defaultdict(dict)
for db in sorted(glob.glob(os.path.join(src, "pass*", "*.db"))):
    d = sqlite3.connect(db)
    for name, counter, calls, avg in d.execute("select kernel_name, counter_name, count(*), avg(value) from "
                                                "counters_collection group by kernel_name, counter_name"):
        data[name][counter] = avg
for name, cs in sorted(data.items()):
    if pat and pat not in name:
        continue
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    print("==", m.group(1) if m else name[:50])
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {v:16.1f}")
