"""Which source files a kernel class is compiled from, and their hashes — so that a counter file under profiles/ can be
tied to the KERNELS it was measured on, not to a commit string.

bench.py pastes roofline.traffic / roofline.binding_bound from profiles/pmc_traffic.json / profiles/sq_bound.json (PMC
counters cannot be collected inside a timed bench run: separate rocprofv3 passes).  Round 4's files carried a commit in a
free-text field only: a kernel edited afterwards would have kept its old counters in the driver's line, silently.  Every
entry now carries the sha256 of the sources of its kernel class (code only: comments and white space do not count) at
collection time; bench.py compares them with the tree
it runs from and prints `"traffic": null, "traffic_stale": true` on any difference (tests/test_bench_model.py edits a
byte of raster.hip and sees exactly that)."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join("godotgaussiansplatting_amd", "csrc")
COMMON = ["gsplat_internal.h"]
# kernel class (gsplat_stats.ms_kernel) -> the files its kernels are compiled from
KERNEL_SOURCES = {
    "project": ["projection.hip", "project_math.h", "sh_eval.h"],
    "scan": ["projection.hip"],
    "emit": ["projection.hip"],
    "splat_sort": ["sort.hip"],
    "sort_upsweep": ["sort.hip"],
    "sort_spine": ["sort.hip"],
    "sort_downsweep": ["sort.hip"],
    "boundaries": ["raster.hip"],
    "render": ["raster.hip", "project_math.h", "sh_eval.h"],
}


def files_of(kernel_class):
    return sorted(set(KERNEL_SOURCES.get(kernel_class, []) + COMMON))


def _code_only(data: bytes) -> bytes:
    """The source without comments and with runs of white space collapsed: what the hash is taken of, so that a comment
    or a re-wrapped line does not declare a kernel changed (a string literal containing '//' would be cut short — there is
    none in these files, and the cut would only ever make two different sources look MORE different)."""
    import re
    text = data.decode("utf-8", "replace")
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip().encode()


def sha_of_tree(root=None, names=None):
    """{file name: sha256 hex} of csrc files as they are on disk under `root` (default: this checkout)."""
    root = root or ROOT
    names = names or sorted({f for v in KERNEL_SOURCES.values() for f in v} | set(COMMON))
    out = {}
    for n in names:
        path = os.path.join(root, CSRC, n)
        out[n] = hashlib.sha256(_code_only(open(path, "rb").read())).hexdigest() if os.path.exists(path) else None
    return out


def sha_of_commit(commit, names=None):
    """The same for a commit of this repository (git show): stamps files collected before hashes were recorded."""
    names = names or sorted({f for v in KERNEL_SOURCES.values() for f in v} | set(COMMON))
    out = {}
    for n in names:
        r = subprocess.run(["git", "-C", ROOT, "show", f"{commit}:{CSRC}/{n}"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        out[n] = hashlib.sha256(_code_only(r.stdout)).hexdigest() if r.returncode == 0 else None
    return out


def stale_files(recorded, kernel_class, root=None):
    """Files of `kernel_class` whose recorded hash differs from the tree's (or is missing): empty list = the counters were
    taken on these very kernels.  recorded: the entry's {file: sha} (None: collected before hashes existed)."""
    names = files_of(kernel_class)
    if not recorded:
        return names
    now = sha_of_tree(root, names)
    return [n for n in names if recorded.get(n) is None or recorded.get(n) != now[n]]


if __name__ == "__main__":
    # python tools/provenance.py stamp <commit>: record the hashes of <commit>'s sources in the counter files whose
    # `_source` names that commit (files collected before this module existed)
    import json
    import sys
    if len(sys.argv) == 3 and sys.argv[1] == "stamp":
        shas = sha_of_commit(sys.argv[2])
        for fn in ("pmc_traffic.json", "sq_bound.json"):
            path = os.path.join(ROOT, "profiles", fn)
            data = json.load(open(path))
            if sys.argv[2] not in data.get("_source", ""):
                print(fn, "was not collected at", sys.argv[2], "- left alone")
                continue
            for cfg, ent in data.items():
                if isinstance(ent, dict) and ent.get("_collected_at", sys.argv[2]) == sys.argv[2]:
                    ent["_csrc_sha256"] = shas
                    ent["_collected_at"] = sys.argv[2]
            json.dump(data, open(path, "w"), indent=1, sort_keys=True)
            print("stamped", fn)
    else:
        print(json.dumps(sha_of_tree(), indent=1))
