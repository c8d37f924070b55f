"""Builds libgsplat_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

-ffp-contract=off and correctly rounded f32 divide/sqrt are part of the arithmetic contract
(DESIGN.md §3): the kernels must round exactly like the CPU oracle.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libgsplat_hip.so")
SOURCES = ["api.hip", "projection.hip", "sort.hip", "raster.hip", "ingest.hip", "group.hip"]
HEADERS = [os.path.join(CSRC, "gsplat_internal.h"), os.path.join(HERE, "..", "include", "gsplat.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        r = subprocess.run([HIPCC, *FLAGS, "-c", path, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}")
        return obj, r.stdout
    return obj, ""


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link the shared library.  Returns its path."""
    if force:
        for s in SOURCES:
            o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if verbose:
        for _, log in results:
            if log.strip():
                print(log)
    if force or _stale(SO, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return SO


def build_test_hooks(out_dir: str) -> str:
    """A TEST build of the library: group.hip compiled with -DGSPLAT_TEST_HOOKS (several members of a one-process group
    may share a device — only the test suite's stand-in for RCCL can serve such a group), linked with the shipped objects
    of every other translation unit.  The shipped library has no such switch."""
    build()
    obj = os.path.join(out_dir, "group_test_hooks.o")
    so = os.path.join(out_dir, "libgsplat_hip_test_hooks.so")
    r = subprocess.run([HIPCC, *FLAGS, "-DGSPLAT_TEST_HOOKS", "-c", os.path.join(CSRC, "group.hip"), "-o", obj],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on group.hip (test hooks):\n{r.stdout}")
    objs = [os.path.join(CSRC, os.path.splitext(s)[0] + ".o") for s in SOURCES if s != "group.hip"] + [obj]
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed (test hooks):\n{r.stdout}")
    return so


def build_variant(name: str, sources, extra_flags, out_dir=None) -> str:
    """An A/B build: `sources` recompiled with `extra_flags`, linked with the shipped objects of the other translation
    units into <out_dir>/libgsplat_<name>.so (GSPLAT_LIB selects it; tools/ab_quick.py).  build_variants/ is scratch."""
    build()
    out_dir = out_dir or os.path.join(HERE, "..", "build_variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        if src in sources:
            obj = os.path.join(out_dir, f"{os.path.splitext(src)[0]}_{name}.o")
            r = subprocess.run([HIPCC, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src} ({name}):\n{r.stdout}")
            objs.append(obj)
        else:
            objs.append(os.path.join(CSRC, os.path.splitext(src)[0] + ".o"))
    so = os.path.join(out_dir, f"libgsplat_{name}.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed ({name}):\n{r.stdout}")
    return so


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 3 and sys.argv[1] == "variant":   # python -m ...build variant <name> <a.hip,b.hip> [-DX=1 ...]
        print(build_variant(sys.argv[2], sys.argv[3].split(","), sys.argv[4:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
