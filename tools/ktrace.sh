#!/bin/bash
# Development helper: per-dispatch durations of the two loop kernels over one render.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt && mkdir -p $R/gpurun_out/kt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -o k -- $R/raytracingrenderer_amd/rtr_main 784 784 ${1:-256} /tmp/img > $R/gpurun_out/kt/log.txt 2>&1 || { tail -5 $R/gpurun_out/kt/log.txt; exit 1; }
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/kt/k_kernel_trace.csv"))]
for name in ("k_trace_queue","k_logic"):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if name in r["Kernel_Name"]]
    n=len(d); print(name, n, "total ms %.1f"%(sum(d)/1e3))
    step=max(1,n//16)
    print("  us by position:", " ".join("%.0f"%(sum(d[i:i+step])/len(d[i:i+step])) for i in range(0,n,step)))
t0=min(int(r["Start_Timestamp"]) for r in rows); t1=max(int(r["End_Timestamp"]) for r in rows)
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
print("span ms %.1f busy ms %.1f"%((t1-t0)/1e6,busy/1e6))
PY
