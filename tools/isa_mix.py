#!/usr/bin/env python3
"""Static instruction mix of one kernel in the compiler's assembly listing, by issue-cost class.

usage: isa_mix.py rtr_api.s <substring of the kernel's symbol> [--blocks]

Classes follow tools/ubench/issue_cycles.hip (cycles per wave-instruction and SIMD at 4 waves per SIMD):
plain fp32/int VALU on vector registers, VALU with a scalar-register source, compares, packed fp32,
transcendental, fp64 and conversions, integer multiplies, max/min.  With --blocks the mix is printed per
basic block (label), which is how the leaf loop of the fused kernel is read.
"""
import re
import sys
from collections import Counter, OrderedDict

COST = {  # cycles per instruction and SIMD, measured (profiles/r03/issue_cycles.jsonl)
    "plain": 2.3, "sgpr_src": 4.2, "cmp": 4.0, "cndmask": 2.3, "packed": 4.2, "trans": 8.1, "f64": 4.2,
    "imul": 4.2, "maxmin": 4.1, "lane": 4.0, "other_valu": 2.3,
}


def classify(op, args):
    if not op.startswith("v_"):
        if op.startswith("s_"):
            return "salu" if not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier")) else \
                ("smem" if op.startswith(("s_load", "s_buffer")) else "sctl")
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        return "misc"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")):
        return "trans"
    if "_f64" in op:
        return "f64"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_hi_i32", "v_mad_i64_i32")):
        return "imul"
    if op.startswith(("v_cmp_", "v_cmpx_")):
        return "cmp"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    srcs = args.split(",")[1:] if "," in args else []
    if op.startswith("v_cndmask"):
        data = srcs[:2]
        if any(re.match(r"\s*-?\|?s(\d+|\[)", a) for a in data):
            return "sgpr_src"
        return "cndmask"
    if op.startswith(("v_max_f32", "v_min_f32", "v_max3", "v_min3", "v_med3")):
        if any(re.match(r"\s*-?\|?s(\d+|\[)", a) for a in srcs):
            return "sgpr_src"
        return "maxmin"
    if any(re.match(r"\s*-?\|?s(\d+|\[)", a) for a in srcs):
        return "sgpr_src"
    return "plain"


def main():
    path, key = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)(;.*)?$", l)
        if not m or l.lstrip().startswith((";", ".")):
            continue
        blocks[cur][classify(m.group(1), m.group(2))] += 1
    total = Counter()
    for b in blocks.values():
        total.update(b)

    def show(name, c):
        valu = sum(v for k, v in c.items() if k in COST)
        cyc = sum(COST[k] * v for k, v in c.items() if k in COST)
        if valu == 0 and not c:
            return
        parts = " ".join("%s=%d" % (k, v) for k, v in sorted(c.items(), key=lambda kv: -kv[1]))
        print("%-12s VALU %5d  cost %7.0f cyc (%.2f per VALU)  | %s" % (name, valu, cyc, cyc / max(valu, 1), parts))

    if per_block:
        for n, c in blocks.items():
            if sum(c.values()) >= 8:
                show(n, c)
    show("TOTAL", total)


if __name__ == "__main__":
    main()
