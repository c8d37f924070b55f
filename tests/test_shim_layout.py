"""The Godot-side shim (shim/) against include/gsplat.h — compile-only checks, no GPU, no Godot.

* the C# P/Invoke structs (shim/GsplatNative.cs) have the header's fields in the header's order with the same C types,
  hence the same offsets and sizes — verified against offsets printed by a C program compiled from the header itself
  (so the test's own parser of the header is checked too) and against the ctypes structures the Python host uses;
* every function the header declares is bound in the C# file (same name, same number of parameters) and exported by the
  library's ctypes EXPORTS list;
* the Godot-free C++ core of the GDExtension (shim/gsplat_bridge.cpp) compiles against the header and links against
  libgsplat_hip.so.
"""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from godotgaussiansplatting_amd import _lib

HEADER = os.path.join(ROOT, "include", "gsplat.h")
CSHARP = os.path.join(ROOT, "shim", "GsplatNative.cs")

C_SIZES = {"uint32_t": 4, "int32_t": 4, "uint64_t": 8, "float": 4, "void *": 8}
CS_TO_C = {"uint": "uint32_t", "int": "int32_t", "ulong": "uint64_t", "float": "float", "IntPtr": "void *"}
ENUM_CONSTANTS = {"GSPLAT_KERNEL_CLASSES": 9}


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def header_structs():
    """{struct name: [(field, c type, array length)]} parsed from include/gsplat.h."""
    text = _strip_comments(open(HEADER).read())
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(void \*|\w+)\s*(.*)$", decl)
            ctype, names = m.group(1), m.group(2)
            for item in names.split(","):
                item = item.strip()
                am = re.match(r"(\w+)\[(\w+)\]$", item)
                if am:
                    n = am.group(2)
                    fields.append((am.group(1), ctype, int(n) if n.isdigit() else ENUM_CONSTANTS[n]))
                else:
                    fields.append((item, ctype, 1))
        out[name] = fields
    return out


def layout(fields):
    """C layout rules (natural alignment): [(name, offset, size)], total size."""
    off, align_max, rows = 0, 1, []
    for name, ctype, count in fields:
        size = C_SIZES[ctype]
        off = (off + size - 1) // size * size
        rows.append((name, off, size * count))
        off += size * count
        align_max = max(align_max, size)
    return rows, (off + align_max - 1) // align_max * align_max


def csharp_structs():
    text = re.sub(r"//.*", "", open(CSHARP).read())
    out = {}
    for name, body in re.findall(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public struct (\w+)\s*\{(.*?)\n    \}", text, flags=re.S):
        fields = []
        for m in re.finditer(r"(?:\[MarshalAs\(UnmanagedType\.ByValArray, SizeConst = (\d+)\)\]\s*)?public (\w+)(\[\])? (\w+);", body):
            count, cstype, is_array, fname = m.groups()
            assert bool(count) == bool(is_array), f"{name}.{fname}: arrays need ByValArray + SizeConst"
            fields.append((fname, CS_TO_C[cstype], int(count) if count else 1))
        out[name] = fields
    return out


def header_functions():
    text = _strip_comments(open(HEADER).read())
    text = text[text.index("typedef struct gsplat_ctx gsplat_ctx;"):]
    funcs = {}
    for ret, name, args in re.findall(r"(int|uint32_t|const char \*)\s*(gsplat_\w+)\(([^)]*)\);", text):
        args = args.strip()
        funcs[name] = 0 if args in ("", "void") else len(args.split(","))
    return funcs


def csharp_functions():
    text = open(CSHARP).read()
    funcs = {}
    for name, args in re.findall(r"\[DllImport\(Lib\)\] public static extern \w+ (gsplat_\w+)\(([^)]*)\);", text):
        funcs[name] = 0 if not args.strip() else len(args.split(","))
    return funcs


PAIRS = {"gsplat_config": ("GsplatConfig", _lib.Config), "gsplat_frame": ("GsplatFrame", _lib.Frame),
         "gsplat_stats": ("GsplatStats", _lib.Stats)}


def test_header_parser_agrees_with_the_compiler(tmp_path):
    structs = header_structs()
    assert set(structs) == set(PAIRS)
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{HEADER}"', "int main(void) {"]
    for sname, fields in structs.items():
        lines.append(f'printf("{sname} %zu\\n", sizeof({sname}));')
        for fname, _, _ in fields:
            lines.append(f'printf("{sname}.{fname} %zu\\n", offsetof({sname}, {fname}));')
    lines += ["return 0;", "}"]
    src = tmp_path / "offsets.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "offsets"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for sname, fields in structs.items():
        rows, size = layout(fields)
        assert int(got[sname]) == size
        for fname, off, _ in rows:
            assert int(got[f"{sname}.{fname}"]) == off, (sname, fname)


@pytest.mark.parametrize("cname", sorted(PAIRS))
def test_csharp_and_ctypes_structs_match_the_header(cname):
    cs_name, ct = PAIRS[cname]
    h_fields = header_structs()[cname]
    cs_fields = csharp_structs()[cs_name]
    assert [f[0] for f in cs_fields] == [f[0] for f in h_fields], "field names / order"
    assert [(f[1], f[2]) for f in cs_fields] == [(f[1], f[2]) for f in h_fields], "field types / array lengths"
    rows, size = layout(h_fields)
    assert layout(cs_fields) == (rows, size)
    assert C.sizeof(ct) == size
    assert [n for n, _ in ct._fields_] == [f[0] for f in h_fields]
    for fname, off, nbytes in rows:
        d = getattr(ct, fname)
        assert (d.offset, d.size) == (off, nbytes), (cname, fname)


def test_every_header_function_is_bound():
    hf, cf = header_functions(), csharp_functions()
    assert len(hf) >= 22
    assert set(hf) == set(_lib.EXPORTS), "ctypes EXPORTS"
    assert set(hf) == set(cf), f"C# DllImport list: missing {set(hf) - set(cf)}, extra {set(cf) - set(hf)}"
    for name, nargs in hf.items():
        assert cf[name] == nargs, f"{name}: {cf[name]} parameters in C#, {nargs} in gsplat.h"
    cs = open(CSHARP).read()
    flags = dict(re.findall(r"#define GSPLAT_FLAG_(\w+) (0x[0-9a-fA-F]+)u", open(HEADER).read()))
    names = {"TIMING": "Timing", "FIX_LAST_TILE": "FixLastTile", "FAST_EXP": "FastExp", "KEEP_EMITTED": "KeepEmitted",
             "KERNEL_TIMING": "KernelTiming", "BLOCK_CULL": "BlockCull", "TIES_STORAGE_ORDER": "TiesStorageOrder", "READBACK_RGB": "ReadbackRgb"}
    for k, v in flags.items():
        assert re.search(rf"public const uint {names[k]} = {v};", cs), k


def test_gdextension_core_compiles_and_links(tmp_path):
    shim = os.path.join(ROOT, "shim")
    lib_dir = os.path.join(ROOT, "godotgaussiansplatting_amd")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                    os.path.join(shim, "gsplat_gdextension.cpp")], check=True)   # empty without godot-cpp, must still parse
    main = tmp_path / "main.cpp"
    main.write_text('#include "%s/gsplat_bridge.h"\nint main() { gsplat_shim::Bridge b(nullptr, 0, 64, 64); '
                    'return (int)b.tile_dims_x() - 4; }\n' % shim)
    exe = tmp_path / "bridge_check"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", str(exe), str(main),
                    os.path.join(shim, "gsplat_bridge.cpp"), "-L" + lib_dir, "-lgsplat_hip", "-lpthread",
                    "-Wl,-rpath," + lib_dir], check=True)
    assert os.path.exists(exe)


def test_gdextension_class_compiles_against_stand_in_godot_cpp(tmp_path):
    """shim/gsplat_gdextension.cpp — the class GDScript would see — parsed and type-checked (-Wall -Wextra -Werror) against
    stand-in declarations of the godot-cpp API it touches (tests/native/godot_cpp_standin: NOT godot-cpp, which is not in
    the image), and linked with its driver.  The GPU suite runs that driver through a session
    (test_gdextension_class_runs_a_session_on_stand_in_godot_cpp)."""
    lib_dir = os.path.join(ROOT, "godotgaussiansplatting_amd")
    standin = os.path.join(ROOT, "tests", "native", "godot_cpp_standin")
    exe = tmp_path / "gdext_driver"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + standin, "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "gdext_driver.cpp"), os.path.join(ROOT, "shim", "gsplat_bridge.cpp"),
                    "-L" + lib_dir, "-lgsplat_hip", "-lpthread", "-Wl,-rpath," + lib_dir], check=True)
    syms = subprocess.run(["nm", "-C", str(exe)], capture_output=True, text=True, check=True).stdout
    assert "gsplat_library_init" in syms and "GsplatBridge::rasterize" in syms
