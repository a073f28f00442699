#!/bin/bash
# Development helper: one rocprofv3 --pmc pass over a Python helper, per-kernel averages.
# usage: tools/pmc_py.sh "<counters>" <kernel-name substring> <script> [args...]
set -o pipefail
R=$GRAFT_REPO_ROOT
CNT=$1; KER=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc && mkdir -p $R/gpurun_out/pmc
( cd $R && timeout -s KILL 280 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc -o p -- python3 "$@" > $R/gpurun_out/pmc/log.txt 2>&1 ) || { tail -5 $R/gpurun_out/pmc/log.txt; exit 1; }
tail -3 $R/gpurun_out/pmc/log.txt
python3 - <<PY
import csv,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open("$R/gpurun_out/pmc/p_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0]
    if "$KER" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg:
    print(k, len(n[k]))
    for c,v in sorted(agg[k].items()): print("   %-26s %.5g" % (c, v/len(n[k])))
PY
