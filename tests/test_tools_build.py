"""The measurement tools that are HIP programs (tools/*.hip) must keep compiling for gfx950: they are built by hand
(`hipcc --offload-arch=gfx950 -O2`), travel to the GPU box as binaries and back the numbers in profiles/r03_valu_rates.md
and r03_step_rates.md.  hipcc cross-compiles without a GPU; nothing is run here."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("tool", ["valu_rates", "step_rates", "pmc_calibrate"])
def test_hip_tools_compile_for_gfx950(tool, tmp_path):
    src = os.path.join(ROOT, "tools", tool + ".hip")
    out = str(tmp_path / tool)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-o", out, src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.getsize(out) > 10000
    # the inline assembly of step_rates is the blend loop's instruction sequences: it must still assemble
    if tool == "step_rates":
        asm = str(tmp_path / "s.s")
        dis = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-S", "--cuda-device-only", "-o", asm, src],
                             capture_output=True, text=True, timeout=600)
        assert dis.returncode == 0, dis.stderr
        text = open(asm).read()
        for needle in ("v_cmpx_lt_f32", "s_and_saveexec_b64", "v_fmaak_f32", "v_lshl_add_u32"):
            assert needle in text, needle


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_stand_in_rccl_compiles_and_exports_what_the_library_binds(tmp_path):
    """tests/native/fake_rccl.hip (the test double the GPU suite hands to gsplat_group_* through GSPLAT_RCCL_LIB) must keep
    compiling, warning-free, and must export every entry point csrc/group.hip resolves with dlsym."""
    import re
    out = str(tmp_path / "libfake_rccl_oneproc.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
                        "-o", out, os.path.join(ROOT, "tests", "native", "fake_rccl.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exported = subprocess.run(["nm", "-D", "--defined-only", out], capture_output=True, text=True, check=True).stdout
    group_src = open(os.path.join(ROOT, "godotgaussiansplatting_amd", "csrc", "group.hip")).read()
    wanted = re.findall(r'GSPLAT_SYM\(\w+, "(nccl\w+)"\)', group_src)
    assert len(wanted) >= 11
    for name in wanted:
        assert re.search(rf"\bT {name}\b", exported), name
