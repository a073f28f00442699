#!/bin/bash
# Development helper: one rocprofv3 --pmc pass over a short render, per-kernel averages.
# usage: tools_pmc.sh "<counters>" [spp]
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc && mkdir -p $R/gpurun_out/pmc
timeout -s KILL 120 rocprofv3 --pmc $1 --output-format csv -d $R/gpurun_out/pmc -o p -- $R/raytracingrenderer_amd/rtr_main 784 784 ${2:-64} /tmp/img > $R/gpurun_out/pmc/log.txt 2>&1 || { tail -5 $R/gpurun_out/pmc/log.txt; exit 1; }
python3 - <<PY
import csv,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open("$R/gpurun_out/pmc/p_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0]
    if "k_trace_queue" not in k and "k_logic" not in k and "k_render_fused" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg:
    print(k, len(n[k]))
    for c,v in sorted(agg[k].items()): print("   %-26s %.5g" % (c, v/len(n[k])))
PY
