#!/bin/bash
# Kernel trace of the wavefront loop on the 1 M-triangle scene (tools/mesh_render_once.py):
# per-kernel statistics, and from the per-dispatch time stamps how much of the render's
# span the GPU spent in which kernel and how much of it nothing ran.
# usage (on the GPU box): tools/profile_mesh_render.sh <label> [spp] [streams]
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mesh_render_${1:-r02}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o mesh -- python3 $R/tools/mesh_render_once.py ${2:-64} 1 ${3:-2} > $OUT/render.txt 2> $OUT/ktrace.log || { tail -5 $OUT/ktrace.log; exit 1; }
timeout -s KILL 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/tools/mesh_render_once.py ${2:-64} 1 ${3:-2} > /dev/null 2> $OUT/pmc_fetch.log || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/tools/mesh_render_once.py ${2:-64} 1 ${3:-2} > /dev/null 2> $OUT/pmc_write.log || { tail -5 $OUT/pmc_write.log; exit 1; }
timeout -s KILL 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o p -- python3 $R/tools/mesh_render_once.py ${2:-64} 1 ${3:-2} > /dev/null 2> $OUT/pmc_sq.log || { tail -5 $OUT/pmc_sq.log; exit 1; }
python3 - <<PY
import csv, glob, os, collections, json
out = "$OUT"
# vector-unit counters per kernel, summed over the program's dispatches
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
        sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump({k: dict(v) for k, v in sq.items() if v.get("SQ_INSTS_VALU", 0) > 1e7}, open(os.path.join(out, "mesh_render_sq.json"), "w"), indent=1, sort_keys=True)
# memory-side traffic per kernel over the whole program (warm-up render included):
# FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes
tr = collections.defaultdict(lambda: {"read_bytes": 0.0, "write_bytes": 0.0, "dispatches": 0})
for d, key, scale in (("pmc_fetch", "read_bytes", 2048.0), ("pmc_write", "write_bytes", 1024.0)):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
            tr[k][key] += float(r["Counter_Value"]) * scale
            if d == "pmc_fetch": tr[k]["dispatches"] += 1
json.dump(tr, open(os.path.join(out, "mesh_render_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in tr.items() if v["read_bytes"] + v["write_bytes"] > 1e8}, indent=1))
for f in glob.glob(os.path.join(out, "ktrace", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(out, "mesh_render_kernel_stats.csv"), "w").write(open(f).read())
    print(open(f).read())
rows = []
for f in glob.glob(os.path.join(out, "ktrace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
# the timed render: everything after the last k_raygen
start = max(i for i, r in enumerate(rows) if "k_raygen" in r[2])
rows = rows[start:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = collections.defaultdict(int)
ev = []
for a, b, k in rows:
    busy[k] += b - a
    ev += [(a, 1), (b, -1)]
ev.sort()
depth, last, idle, two = 0, t0, 0, 0
for t, d in ev:
    if depth == 0: idle += t - last
    if depth >= 2: two += t - last
    depth += d
    last = t
lines = ["span of the render %.2f ms, no kernel running %.2f ms, two or more kernels running %.2f ms, %d dispatches"
         % ((t1 - t0) / 1e6, idle / 1e6, two / 1e6, len(rows))]
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    lines.append("%-40s %9.2f ms in %d dispatches" % (k.split("::")[-1][:40], v / 1e6, sum(1 for r in rows if r[2] == k)))
tq = [(a, b) for a, b, k in rows if "k_trace_queue" in k]
lines.append("traversal launches, ms: " + " ".join("%.2f" % ((b - a) / 1e6) for a, b in tq))
open(os.path.join(out, "mesh_render_timeline.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
print(open(os.path.join(out, "render.txt")).read())
PY
