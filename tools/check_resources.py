#!/usr/bin/env python3
"""Build gate for the fused sampling kernel's register spills.

Reads the compiler's resource-usage remarks (hipcc -Rpass-analysis=kernel-resource-usage,
`make -C raytracingrenderer_amd/csrc resources`) and fails when k_render_fused spills more
scalar registers than the ceilings below.  Scalar spills go into lanes of a vector register
on gfx950; one build of this kernel with 45 of them lost radiance at depth >= 1 (same rays
traced, wrong image), so growth in that number is treated as a defect, not as noise.
Vector spills go to scratch memory: the kernel with 59 of them moved 160 GB per 784 x 784 x
1024 render where its work units name 5 GB, so the flavour that renders the headline scene
is held at none; the tree-walking flavour's are reported and capped.

usage: check_resources.py <file with the remarks> [--json]
"""
import json
import re
import sys

# kernel-name fragment -> (most SGPR spills allowed, least waves per SIMD, most VGPR spills allowed)
GATES = {
    "k_render_fusedILb1ELb1E": (8, 4, 0),    # scene staged in LDS, bounded arithmetic (the Cornell box)
    "k_render_fusedILb1ELb0E": (8, 4, 0),    # scene staged in LDS
    "k_render_fusedILb0ELb0E": (16, 4, 32),  # tree walked in memory
}


def parse(text):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


def main():
    usage = parse(open(sys.argv[1]).read())
    report, bad = {}, []
    for frag, (max_sgpr_spill, min_waves, max_vgpr_spill) in GATES.items():
        names = [n for n in usage if frag in n]
        if len(names) != 1:
            bad.append("%s: %d kernels match" % (frag, len(names)))
            continue
        u = usage[names[0]]
        rec = {"sgpr_spill": int(u.get("SGPRs Spill", -1)), "vgpr_spill": int(u.get("VGPRs Spill", 0)),
               "vgprs": int(u.get("VGPRs", -1)), "scratch_bytes_per_lane": int(u.get("ScratchSize", 0)),
               "waves_per_simd": int(u.get("Occupancy", -1)), "lds_bytes": int(u.get("LDS Size", -1))}
        report[frag] = rec
        if not (0 <= rec["sgpr_spill"] <= max_sgpr_spill):
            bad.append("%s: %d SGPR spills (ceiling %d)" % (frag, rec["sgpr_spill"], max_sgpr_spill))
        if rec["vgpr_spill"] > max_vgpr_spill:
            bad.append("%s: %d VGPR spills (ceiling %d)" % (frag, rec["vgpr_spill"], max_vgpr_spill))
        if rec["waves_per_simd"] < min_waves:
            bad.append("%s: %d waves per SIMD (at least %d)" % (frag, rec["waves_per_simd"], min_waves))
    print(json.dumps(report, indent=1) if "--json" in sys.argv else
          "\n".join("%s: %s" % kv for kv in report.items()))
    if bad:
        print("RESOURCE GATE FAILED: " + "; ".join(bad), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
