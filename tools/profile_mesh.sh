#!/bin/bash
# Profiles of the large-scene traversal (tools/bench_mesh.py: 16.8 M rays against the
# 1 M-triangle height field): rocprofv3 kernel statistics and, in separate passes, the
# memory-side read traffic and L2 hit counters of the traversal kernel.
# usage (on the GPU box): tools/profile_mesh.sh <label>
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mesh_${1:-r01}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o mesh -- python3 $R/tools/bench_mesh.py > $OUT/bench_mesh.txt 2> $OUT/ktrace.log || { tail -5 $OUT/ktrace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/tools/bench_mesh.py > /dev/null 2> $OUT/pmc_fetch.log || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_l2 -o p -- python3 $R/tools/bench_mesh.py > /dev/null 2> $OUT/pmc_l2.log || { tail -5 $OUT/pmc_l2.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o p -- python3 $R/tools/bench_mesh.py > /dev/null 2> $OUT/pmc_sq.log || { tail -5 $OUT/pmc_sq.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections, os
out = "$OUT"
res = {}
for d in ("pmc_fetch", "pmc_l2", "pmc_sq"):
    agg = collections.defaultdict(float); n = set()
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_trace_api_stream" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    for c, v in agg.items(): res[c] = v / len(n)
# FETCH_SIZE is in KiB and tallies 128-byte requests as 64 bytes on gfx950
res["read_bytes_per_launch"] = 2.0 * res.get("FETCH_SIZE", 0.0) * 1024
res["rays_per_launch"] = 1 << 24
if res.get("SQ_INSTS_VALU"):
    # SQ_ACTIVE_INST_VALU counts quad-cycles; lanes active per vector instruction out of 64
    res["valu_active_cycles_per_instruction"] = 4.0 * res["SQ_ACTIVE_INST_VALU"] / res["SQ_INSTS_VALU"]
    res["lanes_active_frac"] = res["SQ_THREAD_CYCLES_VALU"] / (64.0 * res["SQ_INSTS_VALU"])
    res["valu_instructions_per_ray"] = res["SQ_INSTS_VALU"] / res["rays_per_launch"]
json.dump(res, open(os.path.join(out, "mesh_pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(out, "ktrace", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(out, "mesh_kernel_stats.csv"), "w").write(open(f).read())
    print(open(f).read())
print(open(os.path.join(out, "bench_mesh.txt")).read())
print(json.dumps(res, indent=1))
PY
