#!/usr/bin/env python3
"""Issue account of the fused kernel: what its instruction stream costs at the rates this GPU sustains.

usage: issue_account.py <pmc_per_launch.json of tools/profile_round.sh> <issue_cycles.jsonl of
       tools/ubench/issue_cycles.hip> <kernel time of one launch in ms> [--samples N] > issue_account.json

The counter passes give one launch's wave-level instruction counts by class (SQ_INSTS_VALU and its
ADD/MUL/FMA/TRANS/INT32/INT64/CVT/F64 parts, SQ_INSTS_SALU).  The micro-benchmark gives, in shader cycles
per instruction and SIMD at four waves per SIMD, what each kind of instruction costs to issue.  Their
product over the 1024 SIMDs is a FLOOR for the launch: every class is priced at the cheapest instruction it
can contain (an fp32 multiply with a scalar-register operand, a packed one, a comparison or a minimum cost
about twice the plain rate, but the counters do not tell them apart), and scalar instructions are priced at
zero although a wave that issues one is held up for longer than by a vector instruction.
"""
import json
import sys


def main():
    pmc = json.load(open(sys.argv[1]))
    rows = [json.loads(l) for l in open(sys.argv[2]) if l.strip()]
    ms = float(sys.argv[3])
    samples = 784 * 784 * 1024
    if "--samples" in sys.argv:
        samples = int(sys.argv[sys.argv.index("--samples") + 1])

    def cost(name, chains=8):
        for r in rows:
            if r["op"] == name and r["waves_per_simd"] == 4 and r["chains"] == chains:
                return r["cycles_per_inst_per_simd"]
        raise KeyError(name)

    k = next(v for n, v in pmc["sq"].items() if "k_render_fused" in n)
    c = next(v for n, v in pmc["class"].items() if "k_render_fused" in n)
    w = next(v for n, v in pmc["wait"].items() if "k_render_fused" in n)
    valu = k["SQ_INSTS_VALU"]
    plain = cost("v_fma_f32 (vgpr operands)")
    price = {
        "fp32 add / mul / fma (plain, packed or with a scalar operand)": (c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"], cost("v_mul_f32")),
        "transcendental (rcp, rsq, sqrt)": (c["SQ_INSTS_VALU_TRANS_F32"], cost("v_rcp_f32")),
        "32-bit integer": (c["SQ_INSTS_VALU_INT32"], cost("v_add_u32")),
        "64-bit integer (v_mad_u64_u32)": (c["SQ_INSTS_VALU_INT64"], cost("v_mad_u64_u32")),
        "conversions": (c["SQ_INSTS_VALU_CVT"], cost("v_cvt_f64_f32")),
        "fp64 add / mul / fma": (w["SQ_INSTS_VALU_ADD_F64"] + w["SQ_INSTS_VALU_MUL_F64"] + w["SQ_INSTS_VALU_FMA_F64"], cost("v_fma_f64")),
    }
    counted = sum(n for n, _ in price.values())
    price["everything else (moves, selects, comparisons, minima, bit operations)"] = (valu - counted, cost("v_mov_b32"))
    floor_cycles = sum(n * cy for n, cy in price.values()) / 1024.0
    ghz = k["SQ_WAVE_CYCLES"] * 4.0 / k["SQ_WAVES"] / (ms * 1e6) if k.get("SQ_WAVES") else None
    # the clock of the launch: wave-cycles (quad-cycles x 4) per wave over the launch's duration holds when
    # every wave lives for the whole launch, which the persistent fused kernel's do
    out = {
        "kernel": "k_render_fused", "samples_counted": samples, "kernel_ms": ms,
        "valu_instructions": valu, "salu_instructions": k["SQ_INSTS_SALU"],
        "shader_ghz": ghz,
        "plain_fp32_cycles": plain,
        "classes": {n: {"instructions": v, "cycles_per_instruction_and_simd": cy} for n, (v, cy) in price.items()},
        "floor_cycles_of_this_mix_per_simd": floor_cycles,
        "floor_ms": floor_cycles / (ghz * 1e6) if ghz else None,
        "frac_of_mix_floor": floor_cycles / (ghz * 1e6) / ms if ghz else None,
        "cycles_per_valu_instruction_and_simd": ms * 1e6 * ghz / (valu / 1024.0) if ghz else None,
        "lanes_active_frac": k["SQ_THREAD_CYCLES_VALU"] / (64.0 * valu),
        "valu_busy_frac_of_simd_cycles": 4.0 * k["SQ_ACTIVE_INST_VALU"] / 1024.0 / (ms * 1e6 * ghz) if ghz else None,
        "wait_inst_any_frac_of_wave_cycles": w["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"],
        "wait_any_frac_of_wave_cycles": w["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"],
        "note": "floor = instruction counts by class x the cheapest issue cost in the class, over 1024 SIMDs; "
                "scalar instructions priced at zero",
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
