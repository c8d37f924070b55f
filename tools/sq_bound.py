#!/usr/bin/env python3
"""Which limit binds the compositor: SQ counters of `render_kernel` -> profiles/sq_bound.json (bench.py's
roofline.binding_bound).  Runs on the GPU box:  python tools/sq_bound.py <config> [<commit>]

Two rocprofv3 --pmc passes through tools/pmc_one.sh (eight counters each, kernel trace only), averages per launch of the
config's longest render_kernel variant, summed over the chip:
  valu_issue_frac = SQ_INSTS_VALU / 1024 SIMDs x 2.14 cycles per instruction / launch cycles
                    (2.14 = the blend step's measured 54 cycles per 25.2 VALU instructions, profiles/r03_step_rates.md;
                    launch cycles = GRBM_GUI_ACTIVE / 8 XCDs)
  lds_frac        = SQ_LDS_IDX_ACTIVE / 256 CUs / launch cycles
  waves_per_simd  = 4 x SQ_WAVE_CYCLES (quad-cycles) / 1024 SIMDs / launch cycles
"""
import ast
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ["SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY",
          "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    commit = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("GSPLAT_COMMIT", "?")
    merged = {}
    for counters in PASSES:
        out = subprocess.run(["bash", os.path.join(ROOT, "tools", "pmc_one.sh"), cfg, counters, "render_kernel"],
                             cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        for line in out.splitlines():
            if "render_kernel" in line and "{" in line:
                name, d = line[:line.index("{")].strip(), ast.literal_eval(line[line.index("{"):])
                merged.setdefault(name, {}).update(d)
    if not merged:
        print("no counters collected", file=sys.stderr)
        return 1
    name, c = max(merged.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0))  # the frame's heaviest compositor launch
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    ent = {"kernel": name, "launch_cycles": cycles, "counters": c,
           "valu_issue_frac": c["SQ_INSTS_VALU"] / 1024.0 * 2.14 / cycles,
           "lds_frac": c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cycles,
           "waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / 1024.0 / cycles}
    path = os.path.join(ROOT, "profiles", "sq_bound.json")
    allb = json.load(open(path)) if os.path.exists(path) else {}
    allb.setdefault(cfg, {})["render"] = ent
    import provenance
    allb[cfg]["_csrc_sha256"] = provenance.sha_of_tree()   # the sources these kernels were built from
    allb[cfg]["_collected_at"] = commit
    allb["_source"] = ("tools/sq_bound.py: rocprofv3 --pmc SQ counters of render_kernel (two passes), per launch; "
                       "2.14 cycles per VALU instruction = the blend step's measured mix (profiles/r03_step_rates.md); commit " + commit)
    json.dump(allb, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps({cfg: {k: ent[k] for k in ("valu_issue_frac", "lds_frac", "waves_per_simd", "launch_cycles")}}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
